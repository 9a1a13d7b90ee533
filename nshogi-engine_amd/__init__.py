"""nshogi-engine_amd -- MI355X-native batched NN evaluator for nshogi-engine.

Python is only the test / bench harness here: the product is the C-ABI shared
library ``csrc/libnsg.so`` (``include/nsg.h``) and the C++ adapter
``include/nshogi_engine_amd/infer/hip.h``.  This module binds the C ABI with
ctypes and mirrors the reference's executor interface
(``infer::Infer`` -- /root/reference/src/infer/infer.h:19-32) method for method.

There is deliberately no CPU fallback: importing works anywhere (so the
symbol / host-logic tests can run), but every compute call goes through
``libnsg.so`` and raises ``NsgError`` if the library or a HIP device is missing.
"""
from .capi import (  # noqa: F401
    NsgError,
    Evaluator,
    CpuExecutor,
    extract_bits,
    convert_onnx,
    load_library,
    library_path,
    PRECISION_FP32,
    PRECISION_FP16,
    PRECISION_BF16,
    PRECISION_F16X3,
    PRECISION_F16M8,
    MOVE_INDEX_MAX,
    NUM_SQUARES,
)
from . import weights, synth, dist, onnx_io, teacher, positions  # noqa: F401

__all__ = [
    "NsgError", "Evaluator", "CpuExecutor", "extract_bits", "convert_onnx", "load_library",
    "library_path", "weights", "synth", "dist", "onnx_io", "teacher", "positions", "PRECISION_FP32", "PRECISION_FP16",
    "PRECISION_BF16", "PRECISION_F16X3", "PRECISION_F16M8", "MOVE_INDEX_MAX", "NUM_SQUARES",
]
