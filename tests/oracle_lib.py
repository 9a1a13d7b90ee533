"""ctypes wrapper of oracle/liboracle.so -- the CHECKER used by the tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
this (oracle/oracle.h).  The product (libnsg.so) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None


class _Mt(ctypes.Structure):
    _fields_ = [("mt", ctypes.c_uint64 * 312), ("idx", ctypes.c_int)]


class _Net(ctypes.Structure):
    _fields_ = [("in_channels", ctypes.c_int), ("channels", ctypes.c_int),
                ("blocks", ctypes.c_int), ("policy_channels", ctypes.c_int),
                ("value_channels", ctypes.c_int), ("value_hidden", ctypes.c_int),
                ("bn_eps", ctypes.c_float)] + [(n, ctypes.c_void_p) for n in (
                    "stem_w", "stem_bn", "block_w1", "block_bn1", "block_w2", "block_bn2",
                    "policy_w", "policy_b", "value_w", "value_bn", "fc1_w", "fc1_b",
                    "fc2_w", "fc2_b")]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp = ctypes.c_void_p
        lib.nsg_oracle_extract_bits_nchw.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int]
        lib.nsg_oracle_extract_bits_nhwc.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int]
        lib.nsg_oracle_mt_seed.argtypes = [ctypes.POINTER(_Mt), ctypes.c_uint64]
        lib.nsg_oracle_mt_next.argtypes = [ctypes.POINTER(_Mt)]
        lib.nsg_oracle_mt_next.restype = ctypes.c_uint64
        lib.nsg_oracle_uniform01f.argtypes = [ctypes.POINTER(_Mt)]
        lib.nsg_oracle_uniform01f.restype = ctypes.c_float
        lib.nsg_oracle_random_compute.argtypes = [ctypes.POINTER(_Mt), ctypes.c_size_t, vp, vp, vp]
        lib.nsg_oracle_zero_compute.argtypes = [ctypes.c_size_t, vp, vp, vp]
        lib.nsg_oracle_net_from_blob.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(_Net), vp,
                                                 ctypes.c_size_t]
        lib.nsg_oracle_net_forward.argtypes = [ctypes.POINTER(_Net), vp, ctypes.c_int, vp, vp, vp, vp]
        lib.nsg_oracle_evaluate.argtypes = [ctypes.POINTER(_Net), vp, ctypes.c_int, vp, vp, vp]

    # ---- extract bits
    def extract_bits(self, bb, channels_first=True):
        bb = np.ascontiguousarray(bb, dtype=np.uint64)
        b, c = bb.shape[0], bb.shape[1]
        shape = (b, c, 81) if channels_first else (b, 81, c)
        out = np.empty(shape, dtype=np.float32)
        fn = self.lib.nsg_oracle_extract_bits_nchw if channels_first else self.lib.nsg_oracle_extract_bits_nhwc
        fn(out.ctypes.data, bb.ctypes.data, b, c)
        return out

    # ---- executors
    def mt(self, seed):
        st = _Mt()
        self.lib.nsg_oracle_mt_seed(ctypes.byref(st), seed)
        return st

    def mt_next(self, st):
        return int(self.lib.nsg_oracle_mt_next(ctypes.byref(st)))

    def random_compute(self, st, batch):
        p = np.empty((batch, 2187), dtype=np.float32)
        w = np.empty((batch,), dtype=np.float32)
        d = np.empty((batch,), dtype=np.float32)
        self.lib.nsg_oracle_random_compute(ctypes.byref(st), batch, p.ctypes.data, w.ctypes.data,
                                           d.ctypes.data)
        return p, w, d

    def zero_compute(self, batch):
        p = np.full((batch, 2187), np.nan, dtype=np.float32)
        w = np.full((batch,), np.nan, dtype=np.float32)
        d = np.full((batch,), np.nan, dtype=np.float32)
        self.lib.nsg_oracle_zero_compute(batch, p.ctypes.data, w.ctypes.data, d.ctypes.data)
        return p, w, d

    # ---- network
    def net(self, blob):
        return OracleNet(self, blob)


class OracleNet:
    def __init__(self, oracle, blob):
        self.o = oracle
        self.blob = np.frombuffer(bytes(blob), dtype=np.uint8).copy()  # keeps pointers alive
        self.net = _Net()
        self.ptrs = (ctypes.c_void_p * 4096)()
        rc = oracle.lib.nsg_oracle_net_from_blob(self.blob.ctypes.data, self.blob.size,
                                                 ctypes.byref(self.net), self.ptrs, 4096)
        if rc != 0:
            raise ValueError(f"nsg_oracle_net_from_blob failed: {rc}")

    def forward_planes(self, planes, want_trunk=False):
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        b = planes.shape[0]
        p = np.empty((b, 2187), dtype=np.float32)
        v = np.empty((b,), dtype=np.float32)
        d = np.empty((b,), dtype=np.float32)
        t = np.empty((b, self.net.channels, 81), dtype=np.float32) if want_trunk else None
        self.o.lib.nsg_oracle_net_forward(ctypes.byref(self.net), planes.ctypes.data, b,
                                          p.ctypes.data, v.ctypes.data, d.ctypes.data,
                                          t.ctypes.data if want_trunk else None)
        return (p, v, d, t) if want_trunk else (p, v, d)

    def evaluate(self, bb):
        bb = np.ascontiguousarray(bb, dtype=np.uint64)
        b = bb.shape[0]
        p = np.empty((b, 2187), dtype=np.float32)
        v = np.empty((b,), dtype=np.float32)
        d = np.empty((b,), dtype=np.float32)
        self.o.lib.nsg_oracle_evaluate(ctypes.byref(self.net), bb.ctypes.data, b, p.ctypes.data,
                                       v.ctypes.data, d.ctypes.data)
        return p, v, d

    def evaluate_parallel(self, bb, threads=None):
        """evaluate() with the boards spread over host threads (the C call releases the GIL and
        keeps no shared state): what makes a 40-board sample of the 20x256 net a matter of seconds."""
        import threading
        bb = np.ascontiguousarray(bb, dtype=np.uint64)
        b = bb.shape[0]
        threads = max(1, min(b, threads or min(16, os.cpu_count() or 1)))
        outs = [None] * b

        def work(t):
            for i in range(t, b, threads):
                outs[i] = self.evaluate(bb[i:i + 1])

        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return tuple(np.concatenate([o[k] for o in outs]) for k in range(3))


def load():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = Oracle(ctypes.CDLL(path))
    return _lib
