"""The C-ABI library loads and exports every symbol include/nsg.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nsg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nsg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_interface():
    syms = declared_symbols()
    for s in ("nsg_create", "nsg_destroy", "nsg_load", "nsg_compute_nonblocking",
              "nsg_compute_blocking", "nsg_await", "nsg_is_computing", "nsg_reset_gpu",
              "nsg_extract_bits", "nsg_host_register", "nsg_host_unregister"):
        assert s in syms


def test_library_exports_every_declared_symbol(nsg):
    path = nsg.library_path()
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_loads_and_reports_version(nsg):
    lib = nsg.load_library()
    assert lib.nsg_version().decode().startswith("nsg ")


def test_no_cpu_fallback_without_device(nsg):
    """Without a GPU nsg_create must fail loudly, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nsg.NsgError) as e:
        nsg.Evaluator(0, 8)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_never_references_the_oracle():
    """The product tree must not import, link or load anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "nshogi-engine_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", "Makefile")):
                t = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"liboracle|oracle_lib|nsg_oracle_|oracle/", t):
                    bad.append(os.path.join(base, f))
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p) and "oracle" in open(p, errors="replace").read():
            bad.append(p)
    assert not bad, bad
