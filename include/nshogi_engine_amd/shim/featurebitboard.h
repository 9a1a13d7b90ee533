// shim/featurebitboard.h -- stand-in for <nshogi/ml/featurebitboard.h> and
// <nshogi/ml/common.h>, used ONLY when this repository's own host code is
// compiled without libnshogi (the library is not vendored by the reference and
// is absent from this image).  Inside the real engine the
// genuine headers are found first and this file is never included.
//
// It defines exactly what the executor boundary needs: a 16-byte,
// 16-byte-aligned POD of two 64-bit words (layout read off
// /root/reference/src/cuda/extractbit.cu:20-21,47-53) and the two constants
// the boundary is sized by (src/infer/trt.cc:60,62).
#ifndef NSG_SHIM_FEATUREBITBOARD_H
#define NSG_SHIM_FEATUREBITBOARD_H

#include <cstddef>
#include <cstdint>

namespace nshogi {
namespace core {
constexpr std::size_t NumSquares = 81;
} // namespace core
namespace ml {
constexpr std::size_t MoveIndexMax = 27 * core::NumSquares;
struct alignas(16) FeatureBitboard {
    uint64_t Lo; // squares 0..62 in bits 0..62
    uint64_t Hi; // squares 63..80 in bits 0..17, rotate flag bit 24, f32 value bits 63..32
};
static_assert(sizeof(FeatureBitboard) == 16, "FeatureBitboard must be 16 bytes");
} // namespace ml
} // namespace nshogi

#endif
