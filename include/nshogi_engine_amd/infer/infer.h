// infer/infer.h -- the executor plugin interface the engine programs against.
//
// Inside the reference tree this header declares NOTHING: the engine's own
// src/infer/infer.h does.  Either the translation unit has already included it
// (its guard NSHOGI_ENGINE_INFER_INFER_H is then defined -- every file of the
// executor ladders includes "../infer/infer.h" first), or
// -DNSG_USE_REFERENCE_INFER_H together with -Isrc/infer makes this header pull
// it in through the include path.  (The angle form searches the -I directories
// only, never this file's own directory: the quote form that stood here in
// round 2 resolved to THIS file and left infer::Infer undeclared.)
// tests/test_reference_binding.py compiles hip.h / cpu.h / evaluator.h both
// ways against the engine's header.
// Stand-alone, it restates the same abstract class -- same namespace, names,
// signatures and virtual order as /root/reference/src/infer/infer.h:19-32 --
// so the adapter, the Evaluator mirror and the C++ harness in this repository
// compile without the engine.
#ifndef NSG_INFER_INFER_H
#define NSG_INFER_INFER_H

#if defined(NSHOGI_ENGINE_INFER_INFER_H)
// the engine's interface is already in this translation unit
#elif defined(NSG_USE_REFERENCE_INFER_H)
#if !__has_include(<infer.h>)
#error "NSG_USE_REFERENCE_INFER_H needs the engine's src/infer on the include path (-Isrc/infer)"
#endif
#include <infer.h>
#if !defined(NSHOGI_ENGINE_INFER_INFER_H)
#error "<infer.h> on the include path is not nshogi-engine's src/infer/infer.h"
#endif
#else

#if __has_include(<nshogi/ml/featurebitboard.h>)
#include <nshogi/ml/featurebitboard.h>
#include <nshogi/ml/common.h>
#else
#include "../shim/featurebitboard.h"
#endif

#include <cstddef>

#define NSG_INFER_INFER_RESTATED 1

namespace nshogi {
namespace engine {
namespace infer {

class Infer {
 public:
    virtual ~Infer() {
    }
    virtual void computeNonBlocking(const ml::FeatureBitboard* Features,
                                    std::size_t BatchSize, float* DstPolicy,
                                    float* DstWinRate, float* DstDrawRate) = 0;
    virtual void computeBlocking(const ml::FeatureBitboard* Features,
                                 std::size_t BatchSize, float* DstPolicy,
                                 float* DstWinRate, float* DstDrawRate) = 0;
    virtual void await() = 0;
    virtual bool isComputing() = 0;
};

} // namespace infer
} // namespace engine
} // namespace nshogi

#endif // engine header | NSG_USE_REFERENCE_INFER_H | stand-alone
#endif // NSG_INFER_INFER_H
