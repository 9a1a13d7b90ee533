"""The reference-side binding of INTEGRATION.md section 1, compiled against the
engine's OWN interface header (/root/reference/src/infer/infer.h:19-32) with the
exact flags the Makefile branch there prescribes.

Build-container test: it needs /root/reference (absent on the GPU box -> skipped)
and never copies the header; `<nshogi/ml/featurebitboard.h>` / `<nshogi/ml/common.h>`
belong to the absent libnshogi, so the test GENERATES a minimal stand-in for
them in a temporary directory (16-byte POD, MoveIndexMax) -- the boundary needs
nothing else from the library.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INFER = "/root/reference/src/infer"
CSRC = os.path.join(ROOT, "nshogi-engine_amd", "csrc")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_INFER, "infer.h")),
                                reason="the reference tree is not present on this machine")

STUB_FB = """#pragma once
#include <cstddef>
#include <cstdint>
namespace nshogi { namespace ml {
struct alignas(16) FeatureBitboard { uint64_t Lo, Hi; };
} }
"""
STUB_COMMON = """#pragma once
#include <cstddef>
namespace nshogi { namespace core { constexpr std::size_t NumSquares = 81; }
namespace ml { constexpr std::size_t MoveIndexMax = 27 * 81; } }
"""

# what a translation unit of the engine looks like once the `#elif defined(EXECUTOR_HIP)` branch
# of INTEGRATION.md is in (src/mcts/evaluationworker.cc:13-35,76-88)
TU = r"""
%(first)s
#include <nshogi_engine_amd/infer/hip.h>
#include <nshogi_engine_amd/infer/cpu.h>
#include <nshogi_engine_amd/evaluate/evaluator.h>
#include <cstdio>
#include <memory>
#include <type_traits>
#ifndef NSHOGI_ENGINE_INFER_INFER_H
#error "the engine's infer.h was not the one that declared infer::Infer"
#endif
#ifdef NSG_INFER_INFER_RESTATED
#error "the stand-alone restatement of infer::Infer was compiled inside the reference tree"
#endif
using namespace nshogi::engine;
static_assert(std::is_base_of<infer::Infer, infer::Hip>::value, "Hip must derive from the engine's Infer");
static_assert(!std::is_abstract<infer::Hip>::value, "Hip must override all four virtuals");
static_assert(std::is_base_of<infer::Infer, infer::Random>::value && !std::is_abstract<infer::Random>::value, "");
static_assert(std::is_base_of<infer::Infer, infer::Zero>::value && !std::is_abstract<infer::Zero>::value, "");
static_assert(std::is_constructible<infer::Hip, int, uint16_t, uint16_t>::value, "trt.h:44 constructor");
static_assert(sizeof(nshogi::ml::FeatureBitboard) == 16, "");
int main(int Argc, char** Argv) {
    // the ladder of mcts/evaluationworker.cc:76-88 with EXECUTOR=random: one executor + one
    // Evaluator, two batches, results printed for the test to compare with the oracle
    std::unique_ptr<infer::Infer> Infer = std::make_unique<infer::Random>(0);
    evaluate::Evaluator Ev(0, 86, 3, Infer.get(), /*PinMemory*/ false);
    std::FILE* F = std::fopen(Argv[Argc - 1], "wb");
    for (int Batch : {3, 2}) {
        Ev.computeNonBlocking((std::size_t)Batch);
        Ev.await();
        if (Ev.isComputing()) return 2;
        for (int I = 0; I < Batch; ++I) {
            std::fwrite(Ev.getPolicy() + I * nshogi::ml::MoveIndexMax, 4, nshogi::ml::MoveIndexMax, F);
            std::fwrite(Ev.getWinRate() + I, 4, 1, F);
            std::fwrite(Ev.getDrawRate() + I, 4, 1, F);
        }
    }
    std::fclose(F);
    return 0;
}
"""


def _stubs(tmp_path):
    d = tmp_path / "libnshogi_stub" / "nshogi" / "ml"
    d.mkdir(parents=True)
    (d / "featurebitboard.h").write_text(STUB_FB)
    (d / "common.h").write_text(STUB_COMMON)
    return str(tmp_path / "libnshogi_stub")


def _compile(tmp_path, first, defines, link):
    src = tmp_path / "binding_tu.cc"
    src.write_text(TU % {"first": first})
    # CXX_FLAGS += -DEXECUTOR_HIP -DNSG_USE_REFERENCE_INFER_H,
    # INCLUDES += -I$(NSG_DIR)/include -Isrc/infer; the engine builds with -std=c++20 (Makefile:30)
    cmd = ["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-DEXECUTOR_HIP"] + defines + [
        "-I", os.path.join(ROOT, "include"), "-I", REF_INFER, "-I", _stubs(tmp_path), str(src)]
    if link:
        exe = tmp_path / "binding_tu"
        cmd += ["-o", str(exe), "-L", CSRC, "-lnsg", "-Wl,-rpath," + CSRC]
    else:
        exe = None
        cmd += ["-fsyntax-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    return r, exe


def test_adapter_compiles_against_the_engines_infer_h_with_the_documented_flags(tmp_path, nsg, oracle):
    """-DNSG_USE_REFERENCE_INFER_H -Isrc/infer, nothing included first (round 2's shim included itself here)."""
    r, exe = _compile(tmp_path, "", ["-DNSG_USE_REFERENCE_INFER_H"], link=True)
    assert r.returncode == 0, r.stderr[-4000:]
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(str(out), dtype=np.float32).reshape(5, 2189)
    p, w, d = oracle.random_compute(oracle.mt(0), 5)
    np.testing.assert_array_equal(got[:, :2187], p)
    np.testing.assert_array_equal(got[:, 2187], w)
    np.testing.assert_array_equal(got[:, 2188], d)


def test_adapter_compiles_when_the_engine_header_came_first(tmp_path):
    """The executor ladders include "../infer/infer.h" before any executor header; no define needed then."""
    r, _ = _compile(tmp_path, '#include "%s/infer.h"' % REF_INFER, [], link=False)
    assert r.returncode == 0, r.stderr[-4000:]


def test_missing_include_path_is_a_clear_error(tmp_path):
    src = tmp_path / "tu.cc"
    src.write_text("#include <nshogi_engine_amd/infer/hip.h>\nint main() { return 0; }\n")
    r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-DNSG_USE_REFERENCE_INFER_H",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode != 0 and "NSG_USE_REFERENCE_INFER_H needs" in r.stderr
