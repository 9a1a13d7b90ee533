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
#include "bitboard.h"
#include "kernels.h"

namespace nsg {
namespace {

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
// [32 x f16 hi][e2m3(hi) block][e2m3(lo) block], each block 24 B of codes + its E8M0 exponent
// in the dword behind them (kernels.h).  One thread per (square, chunk): it holds the chunk's 32
// values, so the block maxima need no cross-lane step and each block is ONE
// v_cvt_scalef32_pk32_fp6_f16 -- the same rule as the trunk's epilogue (mfma_tile.h: exponent =
// f16 exponent of the block maximum + 110, i.e. the float exponent minus 2; a zero block gets
// 2^-17 and all-zero codes).
constexpr int kThreadsM6 = 384; // 81 squares x 4 chunks = 324 items of the stem's 128 channels in one round

__global__ __launch_bounds__(kThreadsM6) void extractActM6(
    unsigned char* __restrict__ dst, const uint4* __restrict__ src, int channels, int cpad) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) uint4 sBoard[];
    // [channels] bitboards, then the staging image of one round of rows: kThreadsM6 x 8 pieces of 16 bytes.  A thread's
    // row is 128 contiguous bytes of memory, so written from its registers every store instruction touches 64 lines
    // for 16 bytes each (measured 1.5 TB/s, 0.18 of the HBM roof: profiles/r03).  The rows go through LDS instead --
    // piece p of thread t at slot 8 t + (p ^ ((t >> 1) & 7)): sixteen consecutive threads writing one p hit sixteen
    // different 16-byte slots of the 256-byte bank row -- and leave lane-linearly: instruction j of thread u moves piece
    // j * kThreadsM6 + u of the round, 1 KiB of whole lines per wave instruction.
    uint4* stage = sBoard + ((channels + 7) / 8) * 8;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < channels; c += kThreadsM6) sBoard[c] = src[(size_t)b * channels + c];
    __syncthreads();
    const int chunks = cpad / 32;
    const int items = 81 * chunks;
    for (int it0 = 0; it0 < items; it0 += kThreadsM6) {
        const int it = it0 + threadIdx.x;
        if (it < items) {
        const int sq = it / chunks, kc = it - sq * chunks;
        uint32_t hp[16], lp[16];
        u16x2 mh = {0, 0}, ml = {0, 0}; // running maxima of |hi|, |lo| as f16 bit patterns (order-preserving)
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            f32x2 v = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = kc * 32 + 2 * d + q;
                if (c < channels) {
                    const uint4 bb = sBoard[c];
                    const uint64_t l64 = ((uint64_t)bb.y << 32) | bb.x;
                    const uint64_t h64 = ((uint64_t)bb.w << 32) | bb.z;
                    v[q] = __builtin_amdgcn_fmed3f(__uint_as_float(selectBit(l64, h64, sq)), -65000.f, 65000.f);
                }
            }
            const f16x2 h = __builtin_convertvector(v, f16x2);
            const f32x2 lx = v - __builtin_convertvector(h, f32x2);
            hp[d] = __builtin_bit_cast(uint32_t, h);
            lp[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(lx, f16x2));
            mh = __builtin_elementwise_max(mh, __builtin_bit_cast(u16x2, hp[d] & 0x7fff7fffu));
            ml = __builtin_elementwise_max(ml, __builtin_bit_cast(u16x2, lp[d] & 0x7fff7fffu));
        }
        const uint32_t eh = ((uint32_t)(mh[0] > mh[1] ? mh[0] : mh[1]) >> 10) + 110u;
        const uint32_t el = ((uint32_t)(ml[0] > ml[1] ? ml[0] : ml[1]) >> 10) + 110u;
        const u32x16 hsrc = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7],
                             hp[8], hp[9], hp[10], hp[11], hp[12], hp[13], hp[14], hp[15]};
        const u32x16 lsrc = {lp[0], lp[1], lp[2], lp[3], lp[4], lp[5], lp[6], lp[7],
                             lp[8], lp[9], lp[10], lp[11], lp[12], lp[13], lp[14], lp[15]};
        u32x6 hb, lb; // (early clobber: the builtin may place the result inside its source, mfma_tile.h)
        asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(hb) : "v"(hsrc), "v"(__uint_as_float(eh << 23)));
        asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(lb) : "v"(lsrc), "v"(__uint_as_float(el << 23)));
        const int t = threadIdx.x, sw = (t >> 1) & 7;
        uint4* row = stage + t * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) row[i ^ sw] = make_uint4(hp[4 * i], hp[4 * i + 1], hp[4 * i + 2], hp[4 * i + 3]);
        row[4 ^ sw] = make_uint4(hb.s0, hb.s1, hb.s2, hb.s3);
        row[5 ^ sw] = make_uint4(hb.s4, hb.s5, eh, eh); // (the exponent twice: bytes 24 and 28 of the block, kernels.h)
        row[6 ^ sw] = make_uint4(lb.s0, lb.s1, lb.s2, lb.s3);
        row[7 ^ sw] = make_uint4(lb.s4, lb.s5, el, el);
        }
        __syncthreads();
        // the round's rows are contiguous in memory: items it0 .. it0 + n - 1 = n x 128 bytes from there on
        const int n = items - it0 < kThreadsM6 ? items - it0 : kThreadsM6;
        uint4* out = reinterpret_cast<uint4*>(dst + ((size_t)b * 81 * cpad * 4) + (size_t)it0 * 128);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int piece = j * kThreadsM6 + threadIdx.x; // of the round, in memory order
            const int t = piece >> 3, p = piece & 7;
            if (t < n) out[piece] = stage[t * 8 + (p ^ ((t >> 1) & 7))];
        }
        __syncthreads();
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
        const size_t smemM6 = (size_t)((channels + 7) / 8 * 8 + kThreadsM6 * 8) * sizeof(uint4); // bitboards + one round of rows
        hipLaunchKernelGGL(extractActM6, dim3(batch), dim3(kThreadsM6), smemM6,
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
