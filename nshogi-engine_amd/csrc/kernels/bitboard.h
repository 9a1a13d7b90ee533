// bitboard.h -- one square of one feature bitboard (device code shared by extractbit.hip and team_trunk.hip).
//
// The reference's per-element arithmetic, src/cuda/extractbit.cu:15-39: of the 128-bit feature
// word, hi bit 24 = rotate-180 flag, hi bits 63..32 = f32 bit pattern of the plane's value, squares 0..62 = lo bits
// 0..62, squares 63..80 = hi bits 0..17; out = bit(square') ? value : 0 with square' = rotate ? 80 - sq : sq.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace nsg {

__device__ __forceinline__ uint32_t selectBit(uint64_t lo, uint64_t hi, int bit) {
    const uint32_t hi32 = (uint32_t)hi;
    const int rotate = (hi32 >> 24) & 1;
    const uint32_t value = (uint32_t)(hi >> 32);
    const int target = rotate ? 80 - bit : bit;
    const bool useHi = target >= 63;
    const uint64_t word = useHi ? hi : lo;
    const int shift = useHi ? target - 63 : target;
    return ((word >> shift) & 1ULL) ? value : 0u;
}

} // namespace nsg
