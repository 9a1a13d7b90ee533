"""ctypes binding of include/nsg.h (the C-ABI drop-in boundary).

Method names follow the reference's executor interface
(/root/reference/src/infer/infer.h:19-32, src/infer/trt.h:42-58):
computeNonBlocking -> compute_nonblocking, computeBlocking -> compute_blocking,
await -> await_, isComputing -> is_computing, resetGPU -> reset_gpu, load -> load.
"""
import ctypes
import os

import numpy as np

PRECISION_FP32 = 0
PRECISION_FP16 = 1
PRECISION_BF16 = 2
PRECISION_F16X3 = 3
PRECISION_F16M8 = 4
PRECISION_F16M6 = 5
MOVE_INDEX_MAX = 2187
NUM_SQUARES = 81
BITBOARD_BYTES = 16

_PREC_NAMES = {"fp32": 0, "f32": 0, "fp16": 1, "f16": 1, "bf16": 2, "f16x3": 3, "f16m8": 4, "f16m6": 5}


class NsgError(RuntimeError):
    """A C-ABI call returned a negative NSG_E_* code."""

    def __init__(self, code, message):
        super().__init__(f"nsg error {code}: {message}")
        self.code = code


def library_path():
    # NSG_LIB selects an alternate build (e.g. the diagnostic libnsg_diag.so)
    override = os.environ.get("NSG_LIB")
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libnsg.so")


_lib = None


class _Info(ctypes.Structure):
    _fields_ = [
        ("gpu_id", ctypes.c_int), ("batch_size_max", ctypes.c_int),
        ("num_channels", ctypes.c_int), ("channels", ctypes.c_int),
        ("blocks", ctypes.c_int), ("value_channels", ctypes.c_int),
        ("value_hidden", ctypes.c_int), ("precision", ctypes.c_int),
        ("loaded", ctypes.c_int), ("compute_units", ctypes.c_int),
        ("clock_khz", ctypes.c_int), ("param_count", ctypes.c_uint64),
        ("flops_per_position", ctypes.c_double),
        ("trunk_conv_flops_per_position", ctypes.c_double),
        ("device_name", ctypes.c_char * 128),
        ("activation_bound_estimate", ctypes.c_double),
        ("f16m8_window_fallback", ctypes.c_int),
    ]


def load_library():
    """Loads csrc/libnsg.so.  Fails loudly: there is no Python/CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise NsgError(-1, f"{path} is missing: build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'` "
                           "(no fallback path exists)")
    lib = ctypes.CDLL(path)
    vp, sz, fp, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_float), ctypes.c_int
    lib.nsg_last_error.restype = ctypes.c_char_p
    lib.nsg_version.restype = ctypes.c_char_p
    lib.nsg_create.argtypes = [i, i, i, ctypes.POINTER(vp)]
    lib.nsg_destroy.argtypes = [vp]
    lib.nsg_set_precision.argtypes = [vp, i]
    lib.nsg_load.argtypes = [vp, ctypes.c_char_p]
    lib.nsg_convert_onnx.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz)]
    lib.nsg_load_memory.argtypes = [vp, vp, sz]
    lib.nsg_load_device_blob.argtypes = [vp, vp, sz]
    lib.nsg_load_shared.argtypes = [vp, vp]
    for name in ("nsg_compute_nonblocking", "nsg_compute_blocking"):
        getattr(lib, name).argtypes = [vp, vp, sz, vp, vp, vp]
    lib.nsg_await.argtypes = [vp]
    lib.nsg_is_computing.argtypes = [vp]
    lib.nsg_reset_gpu.argtypes = [vp]
    lib.nsg_extract_bits.argtypes = [vp, vp, i, i, i, vp]
    lib.nsg_host_register.argtypes = [vp, sz]
    lib.nsg_host_unregister.argtypes = [vp]
    lib.nsg_upload_features.argtypes = [vp, vp, sz]
    lib.nsg_forward_resident.argtypes = [vp, sz]
    lib.nsg_download_outputs.argtypes = [vp, sz, vp, vp, vp]
    lib.nsg_download_trunk.argtypes = [vp, sz, vp]
    lib.nsg_download_planes_raw.argtypes = [vp, sz, vp, sz, ctypes.POINTER(ctypes.c_size_t)]
    lib.nsg_profile_enable.argtypes = [vp, i]
    lib.nsg_time_planes.argtypes = [vp, sz, i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
    lib.nsg_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_uint64),
                                     ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_uint64)]
    lib.nsg_get_info.argtypes = [vp, ctypes.POINTER(_Info)]
    lib.nsg_get_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    ip = ctypes.POINTER(ctypes.c_int)
    lib.nsg_get_last_plan.argtypes = [vp, ip, ip, ip, ip]
    lib.nsg_get_team_stats.argtypes = [vp, ip, ip, ctypes.POINTER(ctypes.c_uint64)]
    lib.nsg_get_last_launch_kind.argtypes = [vp, ip, ip]
    lib.nsg_get_last_trunk_precision.argtypes = [vp, ip]
    lib.nsg_get_last_split.argtypes = [vp, ip, ip]
    lib.nsg_get_last_slab_split.argtypes = [vp, ip]
    lib.nsg_compute_gather_blocking.argtypes = [vp, vp, sz, vp, vp, i, vp, vp, vp]
    lib.nsg_compute_gather_nonblocking.argtypes = [vp, vp, sz, vp, vp, i, vp, vp, vp]
    lib.nsg_cpu_executor_create.argtypes = [i, ctypes.c_uint64, ctypes.POINTER(vp)]
    lib.nsg_cpu_executor_destroy.argtypes = [vp]
    lib.nsg_cpu_executor_compute.argtypes = [vp, vp, sz, vp, vp, vp]
    _lib = lib
    return lib


def _check(rc):
    if rc < 0:
        raise NsgError(rc, load_library().nsg_last_error().decode("utf-8", "replace"))
    return rc


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _features_array(features, num_channels):
    a = np.ascontiguousarray(features)
    if a.dtype != np.uint64:
        raise TypeError("feature bitboards must be uint64 [batch, channels, 2]")
    if a.ndim != 3 or a.shape[1] != num_channels or a.shape[2] != 2:
        raise ValueError(f"feature bitboards must have shape [batch, {num_channels}, 2]")
    return a


class Evaluator:
    """One (device, stream) executor -- the role of infer::TensorRT
    (src/infer/trt.h:42-88).  Host buffers are caller-owned numpy arrays."""

    def __init__(self, gpu_id, batch_size_max, num_channels=86, precision="fp32"):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.batch_size_max = int(batch_size_max)
        self.num_channels = int(num_channels)
        _check(self._lib.nsg_create(int(gpu_id), int(batch_size_max), int(num_channels),
                                    ctypes.byref(self._h)))
        try:
            prec = _PREC_NAMES[precision] if isinstance(precision, str) else int(precision)
            _check(self._lib.nsg_set_precision(self._h, prec))
        except Exception:
            self.close()  # do not leak the handle (device buffers, streams) on a bad precision
            raise
        self._pending = None

    def close(self):
        if self._h:
            self._lib.nsg_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference interface ------------------------------------------
    def load(self, path):
        _check(self._lib.nsg_load(self._h, os.fsencode(path)))

    def load_memory(self, blob):
        buf = np.frombuffer(blob, dtype=np.uint8)
        _check(self._lib.nsg_load_memory(self._h, _ptr(buf), buf.size))

    def load_device_blob(self, device_ptr, size):
        _check(self._lib.nsg_load_device_blob(self._h, ctypes.c_void_p(int(device_ptr)), int(size)))

    def load_shared(self, src):
        """nsg_load_shared: adopt the network another evaluator has loaded (no file re-read)."""
        _check(self._lib.nsg_load_shared(self._h, src._h))

    def _outputs(self, n, policy, win, draw):
        if policy is None:
            policy = np.empty((n, MOVE_INDEX_MAX), dtype=np.float32)
        if win is None:
            win = np.empty((n,), dtype=np.float32)
        if draw is None:
            draw = np.empty((n,), dtype=np.float32)
        for a, cnt in ((policy, n * MOVE_INDEX_MAX), (win, n), (draw, n)):
            if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"] or a.size < cnt:
                raise ValueError("output buffers must be contiguous float32 of sufficient size")
        return policy, win, draw

    def _batch(self, a, batch_size):
        n = a.shape[0] if batch_size is None else int(batch_size)
        if n < 1 or n > a.shape[0]:
            raise ValueError(f"batch_size {n} outside [1, {a.shape[0]}] (rows of the feature array)")
        return n

    def compute_nonblocking(self, features, batch_size=None, policy=None, win=None, draw=None):
        a = _features_array(features, self.num_channels)
        n = self._batch(a, batch_size)
        policy, win, draw = self._outputs(n, policy, win, draw)
        _check(self._lib.nsg_compute_nonblocking(self._h, _ptr(a), n, _ptr(policy), _ptr(win),
                                                 _ptr(draw)))
        self._pending = (a, policy, win, draw)  # keep caller buffers alive until await
        return policy, win, draw

    def compute_blocking(self, features, batch_size=None, policy=None, win=None, draw=None):
        a = _features_array(features, self.num_channels)
        n = self._batch(a, batch_size)
        policy, win, draw = self._outputs(n, policy, win, draw)
        _check(self._lib.nsg_compute_blocking(self._h, _ptr(a), n, _ptr(policy), _ptr(win),
                                              _ptr(draw)))
        return policy, win, draw

    def compute_gather_blocking(self, features, move_indices, move_offsets, softmax=False,
                                values=None, win=None, draw=None):
        """nsg_compute_gather_blocking: only the logits (or softmax priors) of each position's
        legal moves come back.  move_offsets: batch+1 prefix sums; move_indices: uint16."""
        a = _features_array(features, self.num_channels)
        off = np.ascontiguousarray(move_offsets, dtype=np.uint32)
        idx = np.ascontiguousarray(move_indices, dtype=np.uint16)
        n = off.shape[0] - 1
        total = int(off[-1]) if n >= 0 else 0
        if n < 1 or n > a.shape[0]:
            raise ValueError(f"move_offsets describes {n} positions, the feature array has {a.shape[0]}")
        if idx.size < total:
            raise ValueError(f"move_indices holds {idx.size} entries, move_offsets[-1] = {total}")
        if values is None:
            values = np.full((total,), np.nan, dtype=np.float32)
        if win is None:
            win = np.full((n,), np.nan, dtype=np.float32)
        if draw is None:
            draw = np.full((n,), np.nan, dtype=np.float32)
        for arr, cnt, name in ((values, total, "values"), (win, n, "win"), (draw, n, "draw")):
            if arr.dtype != np.float32 or not arr.flags["C_CONTIGUOUS"] or arr.size < cnt:
                raise ValueError(f"{name} must be contiguous float32 with at least {cnt} elements")
        _check(self._lib.nsg_compute_gather_blocking(self._h, _ptr(a), n, _ptr(idx), _ptr(off),
                                                     1 if softmax else 0, _ptr(values), _ptr(win), _ptr(draw)))
        return values, win, draw

    def await_(self):
        _check(self._lib.nsg_await(self._h))
        self._pending = None

    def is_computing(self):
        return bool(self._lib.nsg_is_computing(self._h))

    def reset_gpu(self):
        _check(self._lib.nsg_reset_gpu(self._h))

    # ---- measurement hooks ----------------------------------------------
    def upload_features(self, features):
        a = _features_array(features, self.num_channels)
        _check(self._lib.nsg_upload_features(self._h, _ptr(a), a.shape[0]))

    def forward_resident(self, batch_size):
        _check(self._lib.nsg_forward_resident(self._h, int(batch_size)))

    def download_outputs(self, batch_size):
        n = int(batch_size)
        policy, win, draw = self._outputs(n, None, None, None)
        _check(self._lib.nsg_download_outputs(self._h, n, _ptr(policy), _ptr(win), _ptr(draw)))
        return policy, win, draw

    def download_trunk(self, batch_size):
        info = self.info()
        out = np.empty((int(batch_size), info["channels"], NUM_SQUARES), dtype=np.float32)
        _check(self._lib.nsg_download_trunk(self._h, int(batch_size), _ptr(out)))
        return out

    def download_planes_raw(self, batch_size):
        """The trunk input of the last forward as raw bytes [batch][81][row_bytes] (debug read-back)."""
        info = self.info()
        cap = int(batch_size) * NUM_SQUARES * 4 * 4 * ((info["num_channels"] + 127) // 128 * 128)
        buf = np.empty(cap, dtype=np.uint8)
        rb = ctypes.c_size_t()
        _check(self._lib.nsg_download_planes_raw(self._h, int(batch_size), _ptr(buf), cap, ctypes.byref(rb)))
        return buf[: int(batch_size) * NUM_SQUARES * rb.value].reshape(int(batch_size), NUM_SQUARES, rb.value)

    def time_planes(self, batch_size, iterations=200):
        """nsg_time_planes: (average ms per launch, algorithmic bytes per launch) of the on-path plane expansion."""
        ms, nbytes = ctypes.c_float(), ctypes.c_double()
        _check(self._lib.nsg_time_planes(self._h, batch_size, iterations, ctypes.byref(ms), ctypes.byref(nbytes)))
        return ms.value, nbytes.value

    def profile_enable(self, enable=True):
        _check(self._lib.nsg_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self):
        t, n = ctypes.c_double(), ctypes.c_uint64()
        f, m = ctypes.c_double(), ctypes.c_uint64()
        _check(self._lib.nsg_profile_read(self._h, ctypes.byref(t), ctypes.byref(n),
                                          ctypes.byref(f), ctypes.byref(m)))
        return {"trunk_ms_total": t.value, "trunk_launches": n.value,
                "forward_ms_total": f.value, "forwards": m.value}

    def info(self):
        s = _Info()
        _check(self._lib.nsg_get_info(self._h, ctypes.byref(s)))
        d = {k: getattr(s, k) for k, _ in _Info._fields_}
        d["device_name"] = s.device_name.decode("utf-8", "replace")
        return d

    def stats(self):
        """nsg_get_stats: forward passes and positions since creation (average batch = ratio)."""
        b, n = ctypes.c_uint64(), ctypes.c_uint64()
        _check(self._lib.nsg_get_stats(self._h, ctypes.byref(b), ctypes.byref(n)))
        return {"batches": b.value, "positions": n.value,
                "average_batch": (n.value / b.value) if b.value else 0.0}

    def team_stats(self):
        """nsg_get_team_stats: is the team trunk in use (1 / 0 / -1 another process holds the device's token),
        workgroups per board of the most recent team launch, launches re-run on the per-layer kernels."""
        en, mem, fb = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64()
        _check(self._lib.nsg_get_team_stats(self._h, ctypes.byref(en), ctypes.byref(mem), ctypes.byref(fb)))
        return {"enabled": en.value, "members_last": mem.value, "fallbacks": fb.value}

    def last_launch_kind(self):
        """nsg_get_last_launch_kind: ("per_layer" | "team" | "coop", cooperative trunk in use 1 / 0 / -1)."""
        k, c = ctypes.c_int(), ctypes.c_int()
        _check(self._lib.nsg_get_last_launch_kind(self._h, ctypes.byref(k), ctypes.byref(c)))
        return {0: "per_layer", 1: "team", 2: "coop"}.get(k.value, k.value), c.value

    def last_plan(self):
        """Launch plan of the most recent forward pass (nsg_get_last_plan)."""
        v = [ctypes.c_int() for _ in range(4)]
        _check(self._lib.nsg_get_last_plan(self._h, *[ctypes.byref(x) for x in v]))
        d = dict(zip(("boards_per_group", "fragments_per_wave", "waves_per_group", "chains"),
                     (x.value for x in v)))
        rs, ks = ctypes.c_int(), ctypes.c_int()
        _check(self._lib.nsg_get_last_split(self._h, ctypes.byref(rs), ctypes.byref(ks)))
        d["row_split"], d["k_split"] = rs.value, ks.value
        ss = ctypes.c_int()
        _check(self._lib.nsg_get_last_slab_split(self._h, ctypes.byref(ss)))
        d["slab_split"] = ss.value
        tp = ctypes.c_int()
        _check(self._lib.nsg_get_last_trunk_precision(self._h, ctypes.byref(tp)))
        d["trunk_precision"] = {v: k for k, v in _PREC_NAMES.items() if k not in ("f32", "f16")}.get(tp.value, tp.value)
        return d


class CpuExecutor:
    """infer::Zero / infer::Nothing / infer::Random (src/infer/{zero,nothing,random}.cc)."""

    KINDS = {"zero": 0, "nothing": 1, "random": 2}

    def __init__(self, kind, seed=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        _check(self._lib.nsg_cpu_executor_create(self.KINDS[kind], int(seed), ctypes.byref(self._h)))

    def compute_blocking(self, batch_size, policy=None, win=None, draw=None):
        n = int(batch_size)
        if policy is None:
            policy = np.full((n, MOVE_INDEX_MAX), np.nan, dtype=np.float32)
        if win is None:
            win = np.full((n,), np.nan, dtype=np.float32)
        if draw is None:
            draw = np.full((n,), np.nan, dtype=np.float32)
        _check(self._lib.nsg_cpu_executor_compute(self._h, None, n, _ptr(policy), _ptr(win),
                                                  _ptr(draw)))
        return policy, win, draw

    def close(self):
        if self._h:
            self._lib.nsg_cpu_executor_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def convert_onnx(data):
    """nsg_convert_onnx: the ONNX -> NSGW v1 conversion nsg_load applies (host only, no device)."""
    lib = load_library()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    need = ctypes.c_size_t()
    _check(lib.nsg_convert_onnx(_ptr(buf), buf.size, None, 0, ctypes.byref(need)))
    out = np.empty(need.value, dtype=np.uint8)
    _check(lib.nsg_convert_onnx(_ptr(buf), buf.size, _ptr(out), out.size, ctypes.byref(need)))
    return out.tobytes()


def extract_bits(dst_ptr, src_ptr, batch_size, num_channels, channels_first=True, stream=0):
    """cuda::extractBits<ChannelsFirst> (src/cuda/extractbit.h:21-23) on DEVICE pointers."""
    lib = load_library()
    _check(lib.nsg_extract_bits(ctypes.c_void_p(int(dst_ptr)), ctypes.c_void_p(int(src_ptr)),
                                int(batch_size), int(num_channels), 1 if channels_first else 0,
                                ctypes.c_void_p(int(stream))))
