// infer/infer.h -- the executor plugin interface the engine programs against.
//
// Inside the reference tree this header is NOT used: src/infer/infer.h is
// (define NSG_USE_REFERENCE_INFER_H and put src/infer on the include path).
// Stand-alone, it restates the same abstract class -- same namespace, names,
// signatures and virtual order as /root/reference/src/infer/infer.h:19-32 --
// so the adapter, the Evaluator mirror and the C++ harness in this repository
// compile without the engine.
#ifndef NSG_INFER_INFER_H
#define NSG_INFER_INFER_H

#if defined(NSG_USE_REFERENCE_INFER_H)
#include "infer.h"
#else

#if __has_include(<nshogi/ml/featurebitboard.h>)
#include <nshogi/ml/featurebitboard.h>
#include <nshogi/ml/common.h>
#else
#include "../shim/featurebitboard.h"
#endif

#include <cstddef>

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

#endif // NSG_USE_REFERENCE_INFER_H
#endif // NSG_INFER_INFER_H
