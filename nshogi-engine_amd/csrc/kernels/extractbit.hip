// extractbit.hip -- feature bitboards -> feature planes, gfx950.
//
// Replaces the reference's only GPU kernel, src/cuda/extractbit.cu
// (K1 :15-39 NCHW, K2 :41-68 NHWC, launcher :76-96).  Same arithmetic per
// output element:
//   hi bit 24 = rotate-180 flag, hi bits 63..32 = f32 bit pattern of the
//   plane's value, squares 0..62 = lo bits 0..62, squares 63..80 = hi bits
//   0..17; out = bit(square') ? value : 0 with square' = rotate ? 80-sq : sq.
// The result is an integer select of a bit pattern, so parity is bit-exact.
//
// The reference launches one 81-thread block per plane (B*C blocks, second
// wave 17/64 occupied, every thread re-reading the same 16 bytes).  Here a
// 256-thread workgroup stages a run of bitboards in LDS with one coalesced
// 16-byte load per lane and then streams the planes out as fully coalesced
// 16-byte stores; the kernel is HBM-write bound (29 240 B per position).
#include "kernels.h"

namespace nsg {
namespace {

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

constexpr int kThreads = 256;

// ---- K1: NCHW.  One workgroup = kPlanes consecutive planes of the flat
// [B*C] plane array = kPlanes*81 consecutive output dwords. ----
constexpr int kPlanes = 64; // 64*81 = 5184 dwords = 1296 16-byte stores

__global__ __launch_bounds__(kThreads) void extractNCHW(
    uint32_t* __restrict__ dst, const uint4* __restrict__ src, int totalPlanes) {
    __shared__ uint4 sPlanes[kPlanes];
    const int plane0 = blockIdx.x * kPlanes;
    const int nPlanes = min(kPlanes, totalPlanes - plane0);
    if (threadIdx.x < nPlanes) {
        sPlanes[threadIdx.x] = src[plane0 + threadIdx.x];
    }
    __syncthreads();

    const int nDwords = nPlanes * 81;
    uint32_t* out = dst + (size_t)plane0 * 81; // 16-byte aligned: 64*81*4
    for (int q = threadIdx.x; q * 4 < nDwords; q += kThreads) {
        const int e0 = q * 4;
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i;
            const int p = min(e / 81, nPlanes - 1);
            const int bit = e - (e / 81) * 81;
            const uint4 bb = sPlanes[p];
            const uint64_t lo = ((uint64_t)bb.y << 32) | bb.x;
            const uint64_t hi = ((uint64_t)bb.w << 32) | bb.z;
            v[i] = selectBit(lo, hi, bit);
        }
        if (e0 + 4 <= nDwords) {
            *reinterpret_cast<uint4*>(out + e0) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            for (int i = 0; e0 + i < nDwords; ++i) out[e0 + i] = v[i];
        }
    }
}

// ---- K2: NHWC, dst[(b*81 + sq)*C + c].  One workgroup per position. ----
__global__ __launch_bounds__(kThreads) void extractNHWC(
    uint32_t* __restrict__ dst, const uint4* __restrict__ src, int channels) {
    extern __shared__ __attribute__((aligned(16))) uint4 sBoard[];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < channels; c += kThreads) {
        sBoard[c] = src[(size_t)b * channels + c];
    }
    __syncthreads();
    const int n = 81 * channels;
    uint32_t* out = dst + (size_t)b * n;
    if ((channels & 1) == 0) {
        // even C: a position's 81*C dwords start 8-byte aligned -> 8-byte stores
        const int half = channels >> 1;
        for (int j = threadIdx.x; j < 81 * half; j += kThreads) {
            const int sq = j / half;
            const int c = (j - sq * half) * 2;
            uint32_t v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 bb = sBoard[c + i];
                const uint64_t lo = ((uint64_t)bb.y << 32) | bb.x;
                const uint64_t hi = ((uint64_t)bb.w << 32) | bb.z;
                v[i] = selectBit(lo, hi, sq);
            }
            *reinterpret_cast<uint2*>(out + (size_t)sq * channels + c) = make_uint2(v[0], v[1]);
        }
        return;
    }
    for (int j = threadIdx.x; j < n; j += kThreads) {
        const int sq = j / channels;
        const int c = j - sq * channels;
        const uint4 bb = sBoard[c];
        const uint64_t lo = ((uint64_t)bb.y << 32) | bb.x;
        const uint64_t hi = ((uint64_t)bb.w << 32) | bb.z;
        out[j] = selectBit(lo, hi, sq);
    }
}

// ---- trunk input: [b][sq][cpad] in the trunk's element type ----
__device__ __forceinline__ uint16_t f32BitsToF16(uint32_t bits) {
    const _Float16 h = (_Float16)__uint_as_float(bits);
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint16_t f32BitsToBf16(uint32_t bits) {
    const __bf16 h = (__bf16)__uint_as_float(bits);
    return __builtin_bit_cast(uint16_t, h);
}

template <int PREC>
__global__ __launch_bounds__(kThreads) void extractAct(
    void* __restrict__ dstv, const uint4* __restrict__ src, int channels, int cpad) {
    extern __shared__ __attribute__((aligned(16))) uint4 sBoard[];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < channels; c += kThreads) {
        sBoard[c] = src[(size_t)b * channels + c];
    }
    __syncthreads();
    const int quadsPerSq = cpad / 4;
    const int nQuads = 81 * quadsPerSq;
    for (int q = threadIdx.x; q < nQuads; q += kThreads) {
        const int sq = q / quadsPerSq;
        const int c0 = (q - sq * quadsPerSq) * 4;
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + i;
            if (c < channels) {
                const uint4 bb = sBoard[c];
                const uint64_t lo = ((uint64_t)bb.y << 32) | bb.x;
                const uint64_t hi = ((uint64_t)bb.w << 32) | bb.z;
                v[i] = selectBit(lo, hi, sq);
            } else {
                v[i] = 0u;
            }
        }
        const uint32_t* vv = v;
        (void)vv;
        const size_t e = ((size_t)b * 81 + sq) * cpad + c0;
        if constexpr (PREC == kF16x3) {
            // row = 128-byte chunks of 32 channels: [32 x f16 hi][32 x f16 lo]
            unsigned char* row = (unsigned char*)dstv + ((size_t)b * 81 + sq) * cpad * 4 +
                                 (size_t)(c0 >> 5) * 128 + (c0 & 31) * 2;
            uint16_t hb[4], lb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fminf(fmaxf(__uint_as_float(vv[i]), -65000.f), 65000.f);
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);
                hb[i] = __builtin_bit_cast(uint16_t, h);
                lb[i] = __builtin_bit_cast(uint16_t, l);
            }
            *reinterpret_cast<uint2*>(row) = make_uint2(hb[0] | ((uint32_t)hb[1] << 16), hb[2] | ((uint32_t)hb[3] << 16));
            *reinterpret_cast<uint2*>(row + 64) = make_uint2(lb[0] | ((uint32_t)lb[1] << 16), lb[2] | ((uint32_t)lb[3] << 16));
        } else if constexpr (PREC == kF16m8) {
            // row = 128-byte chunks of 32 channels: [32 x f16 hi][32 x e4m3(hi)][32 x e4m3(lo * 2^12)]
            unsigned char* row = (unsigned char*)dstv + ((size_t)b * 81 + sq) * cpad * 4 +
                                 (size_t)(c0 >> 5) * 128;
            uint16_t hb[4];
            float hf[4], lf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fminf(fmaxf(__uint_as_float(vv[i]), -65000.f), 65000.f);
                const _Float16 h = (_Float16)v;
                hb[i] = __builtin_bit_cast(uint16_t, h);
                hf[i] = __builtin_amdgcn_fmed3f((float)h, -448.f, 448.f);
                lf[i] = __builtin_amdgcn_fmed3f((v - (float)h) * (float)(1 << kM8LoShift), -448.f, 448.f);
            }
            int h8 = __builtin_amdgcn_cvt_pk_fp8_f32(hf[0], hf[1], 0, false);
            h8 = __builtin_amdgcn_cvt_pk_fp8_f32(hf[2], hf[3], h8, true);
            int l8 = __builtin_amdgcn_cvt_pk_fp8_f32(lf[0], lf[1], 0, false);
            l8 = __builtin_amdgcn_cvt_pk_fp8_f32(lf[2], lf[3], l8, true);
            *reinterpret_cast<uint2*>(row + (c0 & 31) * 2) = make_uint2(hb[0] | ((uint32_t)hb[1] << 16), hb[2] | ((uint32_t)hb[3] << 16));
            *reinterpret_cast<uint32_t*>(row + 64 + (c0 & 31)) = (uint32_t)h8;
            *reinterpret_cast<uint32_t*>(row + 96 + (c0 & 31)) = (uint32_t)l8;
        } else if constexpr (PREC == kFp32) {
            *reinterpret_cast<uint4*>((uint32_t*)dstv + e) =
                make_uint4(v[0], v[1], v[2], v[3]);
        } else if constexpr (PREC == kFp16) {
            uint2 o;
            o.x = f32BitsToF16(v[0]) | ((uint32_t)f32BitsToF16(v[1]) << 16);
            o.y = f32BitsToF16(v[2]) | ((uint32_t)f32BitsToF16(v[3]) << 16);
            *reinterpret_cast<uint2*>((uint16_t*)dstv + e) = o;
        } else {
            uint2 o;
            o.x = f32BitsToBf16(v[0]) | ((uint32_t)f32BitsToBf16(v[1]) << 16);
            o.y = f32BitsToBf16(v[2]) | ((uint32_t)f32BitsToBf16(v[3]) << 16);
            *reinterpret_cast<uint2*>((uint16_t*)dstv + e) = o;
        }
    }
}

// ---- kF16m6 trunk input: per (square, 32-channel chunk) one 128-byte row
// [32 x f16 hi][e2m3(hi) block][e2m3(lo) block], each block 24 B of codes +
// its E8M0 exponent (kernels.h).  One thread per (square, chunk); values are plane bits and four
// scalar planes, so a software encoder is plenty here (the trunk's epilogue uses the packed
// conversion instructions).
__device__ __forceinline__ unsigned encodeE2m3(float q) { // q already divided by the block scale; RNE, saturating
    const unsigned sign = (__float_as_uint(q) >> 31) << 5;
    const float a = fabsf(q);
    if (!(a < 7.5f)) return sign | 0x1fu;
    const int ex = a >= 4.f ? 2 : (a >= 2.f ? 1 : 0);
    const float step = ex == 2 ? 0.5f : (ex == 1 ? 0.25f : 0.125f);
    const float r = rintf(a / step) * step;
    if (r >= 7.5f) return sign | 0x1fu;
    if (r < 1.f) return sign | (unsigned)(r * 8.f);
    const int e2 = r >= 4.f ? 2 : (r >= 2.f ? 1 : 0);
    const float st2 = e2 == 2 ? 0.5f : (e2 == 1 ? 0.25f : 0.125f);
    return sign | (unsigned)(((e2 + 1) << 3) | (((int)(r / st2) - 8) & 7));
}
// 32 values (slot order) -> 24 bytes of codes + exponent byte at [24]
__device__ __forceinline__ void packE2m3BlockDev(const float* v, unsigned char* out) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) m = fmaxf(m, fabsf(v[j]));
    const unsigned ef = __float_as_uint(m) >> 23;          // biased exponent of the maximum (m >= 0)
    const unsigned e8 = m > 0.f ? (ef > 3u ? ef - 2u : 1u) : 0u;
    const float inv = __uint_as_float((254u - e8) << 23);  // 2^(127 - e8)
    uint32_t w[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const unsigned code = e8 ? encodeE2m3(v[j] * inv) : 0u;
        const int bit = 6 * j;
        w[bit / 32] |= code << (bit % 32);
        if (bit % 32 > 26) w[bit / 32 + 1] |= code >> (32 - bit % 32);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) reinterpret_cast<uint32_t*>(out)[i] = w[i];
    reinterpret_cast<uint32_t*>(out)[6] = e8;
    reinterpret_cast<uint32_t*>(out)[7] = 0u;
}

__global__ __launch_bounds__(kThreads) void extractActM6(
    unsigned char* __restrict__ dst, const uint4* __restrict__ src, int channels, int cpad) {
    extern __shared__ __attribute__((aligned(16))) uint4 sBoard[];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < channels; c += kThreads) sBoard[c] = src[(size_t)b * channels + c];
    __syncthreads();
    const int chunks = cpad / 32;
    for (int it = threadIdx.x; it < 81 * chunks; it += kThreads) {
        const int sq = it / chunks, kc = it - sq * chunks;
        float hi[32], lo[32];
        uint32_t hbits[16];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int c = kc * 32 + i;
            float v = 0.f;
            if (c < channels) {
                const uint4 bb = sBoard[c];
                const uint64_t l64 = ((uint64_t)bb.y << 32) | bb.x;
                const uint64_t h64 = ((uint64_t)bb.w << 32) | bb.z;
                v = fminf(fmaxf(__uint_as_float(selectBit(l64, h64, sq)), -65000.f), 65000.f);
            }
            const _Float16 h = (_Float16)v;
            hi[i] = (float)h;
            lo[i] = v - (float)h;
            const uint32_t hb = __builtin_bit_cast(uint16_t, h);
            if (i & 1) hbits[i >> 1] |= hb << 16; else hbits[i >> 1] = hb;
        }
        unsigned char* row = dst + ((size_t)b * 81 + sq) * cpad * 4 + (size_t)kc * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            reinterpret_cast<uint4*>(row)[i] = make_uint4(hbits[4 * i], hbits[4 * i + 1], hbits[4 * i + 2], hbits[4 * i + 3]);
        packE2m3BlockDev(hi, row + 64);
        packE2m3BlockDev(lo, row + 96);
    }
}

// activations [b][sq][c] T -> f32 [b][c][sq]   (debug read-back only)
template <int PREC>
__global__ void actToNCHW(const void* __restrict__ xv, float* __restrict__ dst, int c) {
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < 81 * c; j += blockDim.x) {
        const int ch = j / 81;
        const int sq = j - ch * 81;
        const size_t e = ((size_t)b * 81 + sq) * c + ch;
        float v;
        if constexpr (PREC == kF16x3) {
            const unsigned char* row = (const unsigned char*)xv + ((size_t)b * 81 + sq) * c * 4 +
                                       (size_t)(ch >> 5) * 128 + (ch & 31) * 2;
            v = (float)*(const _Float16*)row + (float)*(const _Float16*)(row + 64);
        } else if constexpr (PREC == kFp32) {
            v = ((const float*)xv)[e];
        } else if constexpr (PREC == kFp16) {
            v = (float)((const _Float16*)xv)[e];
        } else {
            v = (float)((const __bf16*)xv)[e];
        }
        dst[(size_t)b * 81 * c + j] = v;
    }
}

} // namespace

hipError_t launchExtractBitsNCHW(float* dst, const uint64_t* src, int batch,
                                 int channels, hipStream_t stream) {
    const int total = batch * channels;
    if (total <= 0) return hipErrorInvalidValue;
    const int blocks = (total + kPlanes - 1) / kPlanes;
    hipLaunchKernelGGL(extractNCHW, dim3(blocks), dim3(kThreads), 0, stream,
                       (uint32_t*)dst, (const uint4*)src, total);
    return hipGetLastError();
}

hipError_t launchExtractBitsNHWC(float* dst, const uint64_t* src, int batch,
                                 int channels, hipStream_t stream) {
    if (batch <= 0 || channels <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(extractNHWC, dim3(batch), dim3(kThreads),
                       (size_t)channels * sizeof(uint4), stream, (uint32_t*)dst,
                       (const uint4*)src, channels);
    return hipGetLastError();
}

hipError_t launchExtractBitsAct(void* dst, const uint64_t* src, int batch,
                                int channels, int cpad, int prec,
                                hipStream_t stream) {
    if (batch <= 0 || channels <= 0 || cpad % 4 != 0 || cpad < channels)
        return hipErrorInvalidValue;
    const size_t smem = (size_t)channels * sizeof(uint4);
    if (prec == kFp32) {
        hipLaunchKernelGGL(extractAct<kFp32>, dim3(batch), dim3(kThreads), smem,
                           stream, dst, (const uint4*)src, channels, cpad);
    } else if (prec == kFp16) {
        hipLaunchKernelGGL(extractAct<kFp16>, dim3(batch), dim3(kThreads), smem,
                           stream, dst, (const uint4*)src, channels, cpad);
    } else if (prec == kF16x3) {
        hipLaunchKernelGGL(extractAct<kF16x3>, dim3(batch), dim3(kThreads), smem,
                           stream, dst, (const uint4*)src, channels, cpad);
    } else if (prec == kF16m8) {
        hipLaunchKernelGGL(extractAct<kF16m8>, dim3(batch), dim3(kThreads), smem,
                           stream, dst, (const uint4*)src, channels, cpad);
    } else if (prec == kF16m6) {
        if (cpad % 32 != 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL(extractActM6, dim3(batch), dim3(kThreads), smem,
                           stream, (unsigned char*)dst, (const uint4*)src, channels, cpad);
    } else {
        hipLaunchKernelGGL(extractAct<kBf16>, dim3(batch), dim3(kThreads), smem,
                           stream, dst, (const uint4*)src, channels, cpad);
    }
    return hipGetLastError();
}

hipError_t launchActToNCHW(const void* x, float* dst, int batch, int c,
                           int prec, hipStream_t stream) {
    if (prec == kFp32) {
        hipLaunchKernelGGL(actToNCHW<kFp32>, dim3(batch), dim3(256), 0, stream, x, dst, c);
    } else if (prec == kFp16) {
        hipLaunchKernelGGL(actToNCHW<kFp16>, dim3(batch), dim3(256), 0, stream, x, dst, c);
    } else if (prec == kF16x3) {
        hipLaunchKernelGGL(actToNCHW<kF16x3>, dim3(batch), dim3(256), 0, stream, x, dst, c);
    } else {
        hipLaunchKernelGGL(actToNCHW<kBf16>, dim3(batch), dim3(256), 0, stream, x, dst, c);
    }
    return hipGetLastError();
}

} // namespace nsg
