// mfma_tile.h -- the MFMA tile kernel behind every contraction of the
// policy/value/draw network on gfx950 (CDNA4).  Included by the per-precision
// translation units mfma_tile_{fp32,fp16,bf16,f16x3,f16m8}.hip.
//
// It is the body of the opaque TensorRT engine the reference enqueues at
// src/infer/trt.cc:261 (F1 in SURVEY.md 2b; a7 in 8a).  Nothing here is
// translated from the reference -- the reference contains no convolution or
// GEMM code at all.
//
// GEMM view, per workgroup:   D^T[n][m] = sum_k W[k][n] * X[m][k]
//   m = row (a board square, or a board for the value MLP)  -> MFMA column
//   n = output channel                                       -> MFMA row
// Three modes share the main loop:
//   kConv : 3x3 "same" convolution over 9x9 boards, k = (tap, channel).
//   kHeads: 1x1 convolutions of the policy head and the value-feature conv.
//   kDense: plain rows x K GEMM (first layer of the value MLP).
//
// kConv design:
// * Activations live in HBM as [board][square][channel] (channel innermost).
//   A workgroup owns NB whole boards, so every 3x3 neighbour it needs is its
//   own: no halo exchange, and the input tile is staged into LDS ONCE per
//   128-byte channel chunk and then re-read by all nine taps (the 9x im2col
//   blow-up never exists, not even in LDS).
// * LDS image: [8 chunks of 16 B][entry], entry = 24 + b*110 + (y+1)*10 + x.
//   Rows are 10 entries wide: the single zero entry x=9 is both the right
//   halo of row y and the left halo of row y+1; rows y=-1 and y=9 are zero
//   halo rows.  A tap is therefore a constant entry offset dy*10+dx with no
//   bounds test.
// * Every wave owns ALL rows of the tile and NFRAG*16 output channels, so
//   activation fragments are shared through LDS while weight fragments are
//   private to a wave: they stream straight from L2 into registers as 1-KiB
//   fully coalesced wave loads (the host pre-packs them in lane order), two
//   to three K-slabs ahead of use.
// * The MFMA takes the weights as its A operand and the rows as its B
//   operand, so a lane ends up holding 4 consecutive output channels of one
//   row; the channel<->MFMA-row map is chosen at pack time so that a lane's
//   NFRAG fragments form 4*NFRAG consecutive channels: the epilogue reads
//   the residual and writes the result with wide row-contiguous vector
//   accesses and no LDS transpose.
//
// * The epilogue stages each fragment through LDS so that every global access of
//   the residual read and the output write is a full-line, lane-linear 16-byte
//   access, software-pipelined over the fragments.
//
// Precisions: f32 operands -> v_mfma_f32_16x16x4_f32 (exact f32 FMA chain);
// f16/bf16 operands -> v_mfma_f32_16x16x32_{f16,bf16}; kF16x3: split f16 hi/lo,
// three f16 products per MAC; kF16m8: f16 main term + the two correction terms
// on fp8 copies by v_mfma_scale_f32_16x16x128_f8f6f4, over pairs of channel
// chunks (its own main loop below).  Accumulation is f32 in all cases.
#ifndef NSG_MFMA_TILE_H
#define NSG_MFMA_TILE_H

#include "kernels.h"

#include <atomic>
#include <type_traits>

namespace nsg {
namespace tile {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// native vector (not HIP's uint4 struct: struct copies lower to memcpy and
// keep register arrays in scratch)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum Mode { kConv = 0, kHeads = 1, kDense = 2 };

// The cooperative trunk's hand-off: plain stores kept in the XCD's L2, the members of a board on one XCD (coopTrunkKernel)
#ifdef NSG_COOP_SC1_STORES // A/B partner build: write-through stores, no placement requirement
constexpr bool kCoopSameXcd = false;
#else
constexpr bool kCoopSameXcd = true;
#endif
#ifdef NSG_COOP_NO_KEEP // A/B partner build: every chunk of every layer's input through global memory
constexpr bool kCoopKeepOwnSlice = false;
#else
constexpr bool kCoopKeepOwnSlice = true;
#endif
#ifdef NSG_NO_EDGE_ROWS_ONE_BOARD // A/B partner build: one-board MX tiles in natural row order, all taps
constexpr bool kEdgeRowsOneBoard = false;
#else
constexpr bool kEdgeRowsOneBoard = true;
#endif

// SIZE = boards per workgroup (kConv) or 16-row fragments per workgroup.
template <int MODE, int SIZE, int NWAVES, int NBUF = 2>
struct Geom {
    static constexpr bool kBoards = (MODE == kConv);
    static constexpr int kRows = kBoards ? SIZE * 81 : SIZE * 16;
    static constexpr int kMF = (kRows + 15) / 16;
    static constexpr int kEntries = kBoards ? SIZE * 110 + 24 : kMF * 16;
    // Plane p of a chunk buffer starts at p * kPlaneStride = p * (kPlane + 16): one 16-byte slot further on in the banks
    // than the plane before.  A staging store writes, per group of eight lanes, the eight pieces of ONE row (the lane
    // order that makes the global load of a row's 128-byte chunk one cache line per eight lanes): same entry, eight
    // planes -- a multiple of 128 bytes apart they would all fall on the same banks (8-way conflict on every
    // ds_write_b128); one slot apart they cover eight different slots.
    static constexpr int kPlane = (kEntries * 16 + 7 * 16 + 255) / 256 * 256;
    static constexpr int kPlaneStride = kPlane + 16;
    static_assert(7 * kPlaneStride + kEntries * 16 <= 8 * kPlane, "the eight skewed planes must fit the buffer");
    static constexpr int kBuf = 8 * kPlane;  // one 128-byte channel chunk
    static constexpr int kLds = NBUF * kBuf; // double buffered (kF16m8: four buffers = two chunk pairs)
    // LDS the epilogue may stage through: a two-board tile has the CU to itself (its waves need the
    // whole register file), so it takes the whole 160 KiB and moves 9 of its 11 fragments per batch --
    // the residual loads of a batch are one global round trip, and batches run back to back.
    static constexpr int kEpi = kLds; // (the whole 160 KiB = 9 fragments per batch was 3 % slower: fewer, longer batches overlap less)
    // + one trash slot per lane for masked staging lanes (1 KiB) + the row table of the edge-packed tiles (1 KiB)
    static constexpr int kRowTabOff = (kEpi > kLds ? kEpi : kLds) + 1024;
    static constexpr int kLdsAlloc = kRowTabOff + 1024;
    static constexpr int kThreads = NWAVES * 64;
    static constexpr int kItems = (2 * kMF + NWAVES - 1) / NWAVES;
    static constexpr int kTaps = kBoards ? 9 : 1;
};

template <bool BOARDS>
__device__ __forceinline__ int entryOfRow(int m) {
    if constexpr (BOARDS) {
        const int b = m / 81;
        const int sq = m - b * 81;
        const int y = sq / 9;
        const int x = sq - y * 9;
        return 24 + b * 110 + (y + 1) * 10 + x;
    } else {
        return m;
    }
}

// Zeroes the HALO entries of every image plane: the 24 leading entries, per board the rows y = -1 and
// y = 9 and the column x = 9 (nothing reads past the last board's image).  The interior entries are rewritten
// by the staging of every chunk before anything reads them, so clearing the whole image (36 16-byte
// stores per thread for the eight resident buffers of a K-split tile, 2.8k of its 9.7k prologue cycles)
// is 2-3x the work needed.  Lane l owns halo entry l (+64, ...) and walks the planes: one address
// computation per entry, then stores at a constant stride.
// SHIFT (kF16m6 conv images): plane 7 of every chunk buffer -- the second 16-byte piece of the lo block, [code dwords 4-5,
// exponent, exponent] -- lives 8 bytes further on (tileBody, kShiftLo): its halo pieces are cleared there.
template <class G, bool SHIFT = false>
__device__ __forceinline__ void zeroHalo(unsigned char* smem, int tid) {
    static_assert(G::kBoards, "board images only");
    constexpr int kBoardsN = (G::kEntries - 24) / 110;
    constexpr int kHalo = 24 + 29 * kBoardsN;
    constexpr int kPlanes = G::kLds / G::kPlane;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int kWavesN = G::kThreads / 64;
    // Straight-line code: a lane or plane with nothing left to clear writes entry 0 (a leading halo
    // entry, zero anyway) of the last plane again.  The predicated form compiled to one out-of-line
    // block per store at the far end of the kernel: an instruction-cache miss each, in the prologue
    // of every layer.
#pragma unroll
    for (int h0 = 0; h0 < kHalo; h0 += 64) {
        const int h = h0 + lane;
        int e = h; // h < 24: the leading entries
        if (h0 + 63 >= 24) {
            const int q = h - 24, b = q / 29, r = q - b * 29;
            const int within = r < 10 ? r : (r < 19 ? (r - 9) * 10 + 9 : 81 + r); // y = -1 | x = 9 of y = 0..8 | y = 9
            e = h < 24 ? h : 24 + b * 110 + within;
        }
        e = h < kHalo ? e : 0;
#pragma unroll
        for (int pl = 0; pl < (kPlanes + kWavesN - 1) / kWavesN; ++pl) {
            int plane = pl * kWavesN + wave;
            plane = plane < kPlanes ? plane : kPlanes - 1;
            if constexpr (SHIFT) {
                unsigned char* z = smem + e * 16 + (plane >> 3) * G::kBuf + (plane & 7) * G::kPlaneStride + ((plane & 7) == 7 ? 8 : 0);
                *reinterpret_cast<u32x2*>(z) = u32x2{0u, 0u};
                *reinterpret_cast<u32x2*>(z + 8) = u32x2{0u, 0u};
            } else {
                *reinterpret_cast<u32x4*>(smem + e * 16 + (plane >> 3) * G::kBuf + (plane & 7) * G::kPlaneStride) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
}

template <int PREC>
__device__ __forceinline__ void mfmaSlab(f32x4& acc, const u32x4& w, const u32x4& a) {
    if constexpr (PREC == kFp32) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
    } else if constexpr (PREC == kFp16 || PREC == kF16x3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
    }
}

template <int PREC>
__device__ __forceinline__ uint16_t toBits16(float v) {
    if constexpr (PREC == kFp16 || PREC == kF16x3) {
        return __builtin_bit_cast(uint16_t, (_Float16)v);
    } else {
        return __builtin_bit_cast(uint16_t, (__bf16)v);
    }
}
template <int PREC>
__device__ __forceinline__ uint32_t packPair(float lo, float hi) {
    return (uint32_t)toBits16<PREC>(lo) | ((uint32_t)toBits16<PREC>(hi) << 16);
}
template <int PREC>
__device__ __forceinline__ float unpackLo(uint32_t v) {
    if constexpr (PREC == kFp16) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)(v & 0xffffu));
    } else {
        return __uint_as_float(v << 16);
    }
}
// kF16x3: v -> (hi, lo) with hi = f16(v) (clamped to the finite f16 range), lo = f16(v - hi)
__device__ __forceinline__ void splitF16(float v, uint16_t& hi, uint16_t& lo) {
    v = fminf(fmaxf(v, -65000.f), 65000.f);
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, l);
}
// Two values at once: range clamp fused with the ReLU floor (v_med3 takes its inputs unquieted),
// packed conversions (v_cvt_pk_f16_f32) and a packed subtraction.  floorV = 0 (ReLU) or -65000.
__device__ __forceinline__ void splitF16Pair(float a, float b, float floorV, uint32_t& hi, uint32_t& lo) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x2 x = {__builtin_amdgcn_fmed3f(a, floorV, 65000.f), __builtin_amdgcn_fmed3f(b, floorV, 65000.f)};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
// max(x, 0) for the non-NaN values the network produces, as one instruction (fmaxf first quiets its input)
__device__ __forceinline__ float reluF(float x) {
    return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff());
}
__device__ __forceinline__ float f16BitsToF32(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
}

template <int PREC>
__device__ __forceinline__ float unpackHi(uint32_t v) {
    if constexpr (PREC == kFp16) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16));
    } else {
        return __uint_as_float(v & 0xffff0000u);
    }
}

struct Args {
    const unsigned char* x;   // rows x kdim, element type T
    const u32x4* w;           // fragment-ordered weights
    const float* bias;        // [cout]
    const unsigned char* res; // kConv residual, same layout as y (or null)
    unsigned char* y;         // kConv: rows x cout T;  kDense: rows x cout f32
    float* policy;            // kHeads: [board][27*81] f32
    unsigned char* vfeat;     // kHeads: [board][81*VC] T
    int kdim;                 // input channels (multiple of the 128-byte chunk)
    int cout;                 // output channels (multiple of 64)
    int totalRows;            // valid rows (kHeads / kDense)
    int relu;
    int valueChannels;        // kHeads: channels [0,VC) = value conv, [VC,VC+27) = policy
    int vfeatStride;          // kHeads: elements per board row of vfeat (>= 81*VC)
    float accScale;           // accumulators are multiplied by this before the bias (kF16x3: 1/weight scale)
    int outF16x3;             // kF16m8: write the kF16x3 layout (last trunk layer, read by the heads)
    int kSplits;              // kDense: K is split over gridDim.z = kSplits workgroups, each writing its raw
    size_t partStride;        //   partial sums (no bias, no ReLU) at y + z*partStride floats
    unsigned long long* stamps; // diagnostic builds only (NSG_DIAG_STAMPS): 8 u64 per workgroup
    int exp = 0;                // experiment builds only (NSG_EXP_RUNTIME): timing-only switches
};

// What tileKernel takes behind its preloaded scalar arguments.
struct ArgsTail {
    float* policy;
    unsigned char* vfeat;
    int totalRows;
    int valueChannels;
    int vfeatStride;
    int kSplits;
    size_t partStride;
    unsigned long long* stamps;
};
inline ArgsTail tailOf(const Args& a) {
    return ArgsTail{a.policy, a.vfeat, a.totalRows, a.valueChannels, a.vfeatStride, a.kSplits, a.partStride, a.stamps};
}

#ifdef NSG_DIAG_STAMPS
// Stamp 0 (and its wall-clock twin) is held in scalar registers and written with stamp 1: A.stamps is
// not among the preloaded arguments, and the first stamp must not wait for the argument block's tail.
#define NSG_STAMP_DECL unsigned long long nsgStamp0 = 0, nsgReal0 = 0;
#define NSG_STAMP(IDX)                                                                       \
    do {                                                                                     \
        if ((IDX) == 0) {                                                                    \
            nsgStamp0 = __builtin_amdgcn_s_memtime();                                        \
            nsgReal0 = __builtin_amdgcn_s_memrealtime();                                     \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        } else if (A.stamps && threadIdx.x == 0) {                                           \
            unsigned long long* o_ = A.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8; \
            o_[(IDX)] = __builtin_amdgcn_s_memtime();                                        \
            if ((IDX) == 1) { o_[0] = nsgStamp0; o_[7] = nsgReal0; }                         \
            if ((IDX) == 3) o_[6] = __builtin_amdgcn_s_memrealtime();                        \
        }                                                                                    \
    } while (0)
#else
#define NSG_STAMP_DECL
#define NSG_STAMP(IDX) do { } while (0)
#endif

// Single-board conv tiles are small enough for two workgroups per CU (two waves
// per SIMD): the second argument caps registers at 256 so both fit.
template <int MODE, int SIZE, int NWAVES, int NFRAG = 4, int PREC = 0>
constexpr int minWavesPerSimd() {
    // (two boards with 2 fragments per wave would spill at 256 registers: measured slower;
    // the kF16m8 loop needs ~450 registers at any tile size)
    return (MODE == kConv && SIZE == 1 && (NWAVES >= 3 || NFRAG <= 2) && !isMx(PREC)) ? 2 : 1;
}

// kF16m8 slab sequence of one PAIR of channel chunks (A, B): for every tap t the triple
// A.m_t, B.m_t, X_t -- the f16 main terms of the tap in both chunks, then one K=128 MX slab
// with the fp8 correction terms of that tap in both chunks (k-group g: chunk g>>1, term g&1).
template <int TAPS>
struct M8Seq {
    static constexpr int kSlabs = 3 * TAPS;
    static constexpr bool isX(int s) { return s % 3 == 2; }
    static constexpr int tap(int s) { return s / 3; }
    static constexpr int half(int s) { return s % 3; }                 // main slabs: 0 = chunk A, 1 = chunk B
    static constexpr int mainOrd(int s) { return 2 * (s / 3) + s % 3; } // main slabs in stream order
    static constexpr int slabOfMain(int o) { return 3 * (o / 2) + (o & 1); }
    static constexpr int slabOfX(int t) { return 3 * t + 2; }
    // record sets (nft x 1 KiB) before slab s: per triple one, one, two
    static constexpr int recOff(int s) { return 4 * (s / 3) + (s % 3 == 2 ? 2 : s % 3); }
    static constexpr int kRecPair = 4 * TAPS;
};
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

// Row order of a TWO-BOARD MX tile ("edge-packed rows").  A 3x3 "same" convolution multiplies 14 % of its
// (row, tap) pairs by the zero halo: every square of the top row has no dy = -1 neighbours, and so on.  In
// natural row order every 16-row fragment mixes edge and interior squares and needs all nine taps; here
// the 64 edge squares of the two boards form four fragments that each lack one whole line of taps, and the
// MFMAs (and LDS fragment reads) of those (fragment, tap) pairs are simply not issued -- adding w * 0 is
// an exact no-op, so the outputs are bit-identical to the full loop's.  Fragments:
//   0..5  interior squares (y, x in 1..7), 96 of the 98        all nine taps
//   6     top rows, x = 1..8 of both boards                    no dy = -1 taps
//   7     bottom rows, x = 0..7                                no dy = +1 taps
//   8     left columns y = 1..7 + the last 2 interior squares  all nine taps
//   9     right columns y = 1..7 + the corners (8,8)           no dx = +1 taps
//   10    the corners (0,0) (+ 14 padding rows)                the four taps dy >= 0, dx >= 0
// 99 -> 85 (fragment, tap) pairs per chunk: 14 % fewer matrix instructions and fragment reads.  (Round 4: the fragment
// that exists for two rows only used to hold the two interior squares that fit nowhere else -- nine taps for two
// rows; it now holds two corners, four taps, and the interior squares take the corners' places among the left columns,
// whose fragment then runs every tap: 87 -> 85.)
// (CORNERS = false: round 3's assignment -- corners (0,0) in fragment 8, no dx = -1 taps there, the two left-over
// interior squares in fragment 10 with all nine taps, 87 pairs -- which the opt-in slab-split tiles keep: their waves'
// static shares of a chunk pair were balanced for it.)
template <bool CORNERS>
struct EdgeRowsT {
    static constexpr int kMF = 11;
    static constexpr bool needTap(int f, int t) {
        if (!CORNERS) return f == 6 ? t / 3 != 0 : f == 7 ? t / 3 != 2 : f == 8 ? t % 3 != 0 : f == 9 ? t % 3 != 2 : true;
        return f == 6 ? t / 3 != 0 : f == 7 ? t / 3 != 2 : f == 9 ? t % 3 != 2 : f == 10 ? (t / 3 != 0 && t % 3 != 0) : true;
    }
    // board, rank and file of row r of fragment f; false = padding row
    static constexpr __host__ __device__ bool square(int f, int r, int& b, int& y, int& x) {
        b = r >> 3;
        const int k = r & 7;
        if (f == 10 && r >= 2) return false;
        if (CORNERS ? f == 10 : (f == 8 && k == 7)) { // the two (0, 0) corners
            b = CORNERS ? r : r >> 3; y = 0; x = 0;
        } else if (f < 6 || (CORNERS ? (f == 8 && k == 7) : f == 10)) { // (... the two interior squares left over)
            const int i = f < 6 ? f * 16 + r : 96 + (CORNERS ? (r >> 3) : r);
            b = i >= 49 ? 1 : 0;
            const int j = i - b * 49;
            const int q = (j * 37) >> 8; // j / 7 for j < 49
            y = 1 + q;
            x = 1 + j - q * 7;
        } else if (f == 6) {
            y = 0; x = 1 + k;
        } else if (f == 7) {
            y = 8; x = k;
        } else if (f == 8) {
            y = 1 + k; x = 0;
        } else {
            y = k < 7 ? 1 + k : 8; x = 8;
        }
        return true;
    }
    // natural row index b*81 + y*9 + x of row r of fragment f; -1 = padding row
    static constexpr __host__ __device__ int rowOf(int f, int r) {
        int b = 0, y = 0, x = 0;
        return square(f, r, b, y, x) ? b * 81 + y * 9 + x : -1;
    }
};

// The same for ONE board (the one-board persistent tiles of 144 ... 256 boards): 81
// squares in six fragments.  Fragments 0-2: 48 interior squares; 3: the left and right columns (y = 1..7) + the last
// interior square -- all nine taps, and with the first three what every slab starts with (the loop spreads a slab's
// record requests over its first four steps: those four fragments run every tap); 4: the top row with its two corners,
// no dy = -1 taps; 5: the bottom row with its corners, no dy = +1 taps.  54 -> 48 (fragment, tap) pairs per chunk: 11 %
// fewer matrix instructions and fragment reads.
struct EdgeRows1 {
    static constexpr int kMF = 6;
    static constexpr bool needTap(int f, int t) { return f == 4 ? t / 3 != 0 : f == 5 ? t / 3 != 2 : true; }
    static constexpr __host__ __device__ bool square(int f, int r, int& b, int& y, int& x) {
        b = 0;
        if (f < 3 || (f == 3 && r == 14)) { // interior square i
            const int i = f < 3 ? f * 16 + r : 48;
            const int q = (i * 37) >> 8; // i / 7 for i < 49
            y = 1 + q;
            x = 1 + i - q * 7;
            return true;
        }
        if (f == 3) { // left column, right column
            if (r > 14) return false;
            y = 1 + (r < 7 ? r : r - 7);
            x = r < 7 ? 0 : 8;
            return true;
        }
        // top (f == 4) / bottom row: x = 0..8
        if (r > 8) return false;
        y = f == 4 ? 0 : 8;
        x = r;
        return true;
    }
};

// "Slab split" (SS waves per 64-channel group, two-board MX tiles at mid batches): the SS waves of a channel
// group share the K range of every chunk pair by SLABS -- each runs its own static subset of the pair's 27
// slabs (18 f16 main slabs M0..M17 = (tap, chunk A|B), 9 MX slabs X0..X8) for ALL row fragments and the
// group adds its accumulators up at the end.  A two-board tile of 64 channels is then one workgroup:
// at 128 boards 256 workgroups, each streaming a quarter of the layer's weights through its L1 for TWO
// boards (the one-board tiles stream half of them for one), with the two-board tile's edge-packed rows.
// The subsets are balanced in (slab, fragment) steps with edge-packed rows (66 / 67 / 62 / 66 of 261 for
// four parts, 132 / 129 for two), every part's main slabs come in multiples of three (three weight register
// sets rotate) and no part starts a pair with an MX slab.  Entries: main slab o = o, MX slab of tap t = 100 + t.
template <int SS, int PART>
struct OwnSeq {
    struct Tab {
        int n, nMain, nX;
        short slab[27], mainSlab[18], xSlab[9], mainOrd[27], xOrd[27];
    };
    static constexpr int kMaxList = 27;
    static constexpr Tab make() {
        int list[kMaxList] = {};
        int n = 0;
        auto M = [&](int o) { list[n++] = o; };
        auto X = [&](int t) { list[n++] = 100 + t; };
        if (SS == 1) {
            for (int t = 0; t < 9; ++t) { M(2 * t); M(2 * t + 1); X(t); }
        } else if (SS == 2 && PART == 0) {
            X(0); M(0); X(1); M(1); X(2); M(2); X(3); M(3); X(5); M(4); X(6); M(5); X(7); X(8);
        } else if (SS == 2 && PART == 1) {
            for (int o = 6; o < 16; ++o) M(o);
            X(4); M(16); M(17);
        } else if (SS == 4 && PART == 0) {
            M(0); X(0); M(1); X(1); M(2); X(2); X(3);
        } else if (SS == 4 && PART == 1) {
            M(3); X(4); M(4); X(5); M(5); X(6); X(8);
        } else if (SS == 4 && PART == 2) {
            for (int o = 6; o < 12; ++o) M(o);
        } else if (SS == 4 && PART == 3) {
            M(12); M(13); M(14); M(15); X(7); M(16); M(17);
        }
        Tab t{};
        t.n = n;
        for (int u = 0; u < n; ++u) {
            const bool isx = list[u] >= 100;
            const int id = isx ? list[u] - 100 : list[u];
            t.slab[u] = (short)(isx ? 3 * id + 2 : 3 * (id / 2) + (id & 1));
            t.mainOrd[u] = (short)(isx ? -1 : t.nMain);
            t.xOrd[u] = (short)(isx ? t.nX : -1);
            if (isx) t.xSlab[t.nX++] = t.slab[u];
            else t.mainSlab[t.nMain++] = t.slab[u];
        }
        return t;
    }
    static constexpr Tab kTab = make();
    static constexpr int kN = kTab.n, kMain = kTab.nMain, kX = kTab.nX;
    static constexpr int slab(int u) { return kTab.slab[u]; }           // real slab (M8Seq numbering) at own position u
    static constexpr int mainOrd(int u) { return kTab.mainOrd[u]; }     // own ordinal of the main slab at u
    static constexpr int xOrd(int u) { return kTab.xOrd[u]; }           // own ordinal of the MX slab at u
    static constexpr int mainSlab(int o) { return kTab.mainSlab[o]; }
    static constexpr int xSlab(int j) { return kTab.xSlab[j]; }
    // MX records: two register sets.  An even number of own MX slabs runs as one cyclic pipeline across pairs
    // (the next one is requested at the top of the current one); otherwise the pair's first one is requested at
    // the top of the pair (set 0 is free once the previous pair's last MX slab has been issued).
    static constexpr bool kXCyclic = kX >= 2 && kX % 2 == 0;
    static_assert(kMain % 3 == 0, "three f16 weight sets must carry across chunk pairs");
    static_assert(kX == 0 || kTab.xOrd[0] < 0 || kXCyclic, "a pair must not start with an MX slab it requests at its top");
};

// The (own slab, fragment) steps of one chunk pair in issue order, and their inverse: with edge-packed rows
// the steps whose tap a fragment does not need are left out; the sequence is padded with null steps to a
// multiple of the fragment window (the window slot of a step must be the same in every pair).
template <class OS, int MF, bool PERM, int WIN, class ER>
struct StepSeq {
    static constexpr int kMax = (27 * MF + WIN - 1) / WIN * WIN;
    struct Tab {
        int n, padded;
        short pos[kMax], frag[kMax], index[27 * MF], xord[kMax];
    };
    static constexpr bool active(int u, int f) { return !PERM || ER::needTap(f, OS::slab(u) / 3); }
    static constexpr Tab make() {
        Tab t{};
        int n = 0;
        for (int u = 0; u < OS::kN; ++u)
            for (int f = 0; f < MF; ++f) {
                t.index[u * MF + f] = (short)(active(u, f) ? n : -1);
                if (active(u, f)) {
                    t.pos[n] = (short)u;
                    t.frag[n] = (short)f;
                    ++n;
                }
            }
        t.n = n;
        t.padded = (n + WIN - 1) / WIN * WIN;
        for (int q = n; q < kMax; ++q) { t.pos[q] = 0; t.frag[q] = -1; }
        int nx = 0; // MX steps (slab % 3 == 2) in front of step q
        for (int q = 0; q < kMax; ++q) {
            t.xord[q] = (short)nx;
            if (q < n && OS::slab(t.pos[q]) % 3 == 2) ++nx;
        }
        return t;
    }
    static constexpr Tab kTab = make();
    static constexpr int kReal = kTab.n;        // steps that issue MFMAs
    static constexpr int kSteps = kTab.padded;  // ... plus the null steps behind them
    static constexpr int pos(int q) { return kTab.pos[q]; }
    static constexpr int frag(int q) { return kTab.frag[q]; } // -1: null step
    static constexpr int index(int u, int f) { return kTab.index[u * MF + f]; }
    static constexpr int xord(int q) { return kTab.xord[q]; } // ordinal of MX step q among the pair's MX steps
};

// One layer's work for this workgroup.  RES: 0 = no residual, 1 = residual,
// 2 = decided at run time by A.res (persistent trunk kernel).
// MS (row split, small one-board tiles): the waves of a workgroup form MS groups, each computing
// 1/MS of the tile's row fragments for NWAVES/MS channel slices, instead of every wave reading every
// row fragment -- a one-fragment-per-wave tile is LDS-bandwidth-bound, and two fragments per wave on
// half the rows is the same work per wave for half the LDS reads.
// KS (K split, kF16m8 one-board tiles at mid batches): the waves of a channel group each take 1/KS of the
// input-channel chunk pairs for ALL row fragments and sum their accumulators through LDS at the end.
// A row split makes every wave of the group stream the group's whole weight set through the L1 (64 B/clk
// per CU: 2.4 MB = 17 us per layer, more than the MFMAs of a one-board tile take); the K split streams
// it once.  All 2*KS*... chunk tiles of the board are resident in LDS at once (eight image buffers for
// 256 channels), so the loop has no staging and no barriers.
// COOP (coopTrunkKernel): the workgroup's input rows were written by OTHER workgroups of the same launch -- every load
// of activations (input tiles, residual) is an agent-scope (sc1) buffer load that never hits this CU's L1; every output
// store is a plain store that stays in the XCD's L2 (kCoopSameXcd: the members of a board share an XCD, verified at run
// time) or, in the -DNSG_COOP_SC1_STORES build, a write-through (sc1) store.
// KEEP (cooperative trunk, one row group per board): the epilogue builds this workgroup's output rows -- its 64- or
// 128-channel slice of the board -- straight in the NEXT layer's LDS image (the chunk buffers of its own channels: the
// image is the staging area of the row-major move to global memory as well), and the next layer, told so by `haveOwn`,
// neither requests nor stages those chunks: half the tile of a two-way K split, a quarter of a four-way one, never
// leaves the CU.
template <int PREC, int MODE, int SIZE, int NFRAG, int NWAVES, int RES, int MS = 1, int KS = 1, int SS = 1, int PART = 0, bool COOP = false,
          bool KEEP = false>
__device__ __forceinline__ void tileBody(const Args& A, unsigned char* smem, bool zeroLds, bool haveOwn = false) {
    static_assert(!KEEP || (COOP && SIZE == 1 && MS == 1 && KS > 1 && PREC == kF16m6), "own slice kept in LDS: cooperative K-split tiles");
    static_assert(!COOP || (isMx(PREC) && MODE == kConv && NFRAG == 4), "cooperative trunk: MX conv tiles");
    static_assert(SS == 1 || (isMx(PREC) && MODE == kConv && SIZE == 2 && MS == 1 && KS == 1 && NWAVES % SS == 0 && PART < SS),
                  "slab split: two-board MX conv tiles");
    using G = Geom<MODE, SIZE, NWAVES, (isMx(PREC) ? (KS > 1 ? 8 : 4) : 2)>;
    static_assert(KS == 1 || (isMx(PREC) && MODE == kConv && SIZE == 1 && NWAVES % KS == 0),
                  "K split: kF16m8 one-board conv tiles");
    // K split AND row split: the row groups are separate WORKGROUPS (blockIdx.z), each with all its waves on
    // the K parts -- eight workgroups per board for the smallest batches, whose layers are a latency chain
    // (the main loop of a four-way K split is 17k of its 30k cycles: half the rows, half of that)
    constexpr bool kRowWG = (MS > 1 && KS > 1);
    const bool hasRes = (RES == 1) || (RES == 2 && A.res != nullptr);
    NSG_STAMP_DECL
    NSG_STAMP(0);
    constexpr bool kM8 = isMx(PREC);          // f16 main term + MX correction terms (the kF16m8 main loop)
    constexpr bool kM6 = (PREC == kF16m6);    // ... with e2m3 operands and per-block E8M0 scales
    constexpr int ES = (PREC == kFp32 || PREC == kF16x3 || kM8) ? 4 : 2;
    constexpr bool kSplit = (PREC == kF16x3);
    static_assert(!kM8 || (MODE == kConv && NFRAG == 4), "kF16m8: full trunk-conv tiles only");
    static_assert(NFRAG == 1 || NFRAG == 2 || NFRAG == 4, "fragments per wave");
    static_assert(MS == 1 || (MODE == kConv && SIZE == 1 && G::kMF % MS == 0 && (KS > 1 || NWAVES % MS == 0) &&
                              (NFRAG < 4 || isMx(PREC))),
                  "row split: one-board conv tiles (small tiles, or kF16m8 full-channel tiles)");
    constexpr int kMFw = G::kMF / MS; // row fragments this wave computes
    // edge-packed row order (EdgeRows): the two-board MX tiles
#ifdef NSG_NO_EDGE_ROWS // A/B partner build (make ab ABFLAGS=-DNSG_NO_EDGE_ROWS): natural row order, all taps
    constexpr bool kPerm = false;
#else
    // (one board: the tiles that stage their chunks, KS == 1 -- the persistent tiles of 144 ... 256 boards: +3...3.7 %.  On
    // the K-split tiles the table-driven epilogue of a wave's two or three fragments costs more than the skipped taps
    // save: -0.6...-2 % at 33-128 boards, profiles/r04/zz_*; the code paths are there, `&& KS == 1` is the switch.)
    constexpr bool kPerm = isMx(PREC) && MODE == kConv && NFRAG == 4 && MS == 1 && KS == 1 && (SIZE == 2 || (SIZE == 1 && kEdgeRowsOneBoard));
#endif
    using EdgeRows = std::conditional_t<SIZE == 1, EdgeRows1, EdgeRowsT<SS == 1>>;
    static_assert(!kPerm || kMFw == EdgeRows::kMF, "edge-packed rows: one table row per fragment");

    int tidOpaque = threadIdx.x;
    // (opaque to the optimiser: inside the persistent trunk kernel the per-lane address tables
    // below would otherwise be hoisted out of the layer loop and stay live through the epilogue)
    asm volatile("" : "+v"(tidOpaque));
    const int tid = tidOpaque;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int g = lane >> 4;
    const size_t row0 = (size_t)blockIdx.x * G::kRows;
    int nkc = A.kdim * ES / 128;
    int kc0 = 0; // first channel chunk of this workgroup (kDense split-K)
    if constexpr (MODE == kDense) {
        const int per = nkc / A.kSplits; // the host picks a divisor
        kc0 = blockIdx.z * per;
        nkc = per;
    }
    const int nft = A.cout / 16;
    const int waveGroup = (blockIdx.y * NWAVES + wave) / (kRowWG ? KS : MS * KS * SS); // group of NFRAG channel fragments
    const int fBase = (MS > 1) ? (kRowWG ? (int)blockIdx.z : wave % MS) * kMFw : 0; // first row fragment of this wave
    // rows this workgroup may write (a K part past the workgroup's last fragment adds up and converts
    // whatever lies there: with the rows split over workgroups those rows belong to the next one)
    const int rowLimit = kRowWG ? ((fBase + kMFw) * 16 < G::kRows ? (fBase + kMFw) * 16 : G::kRows) : G::kRows;
    // kConv buffers are sized for whole workgroups; flat modes clamp rows.
    const size_t lastRow = G::kBoards ? ~(size_t)0 : (size_t)(A.totalRows - 1);

    // staging: item k of this wave moves 8 rows x 8 pieces (16 B per lane) = the whole 128-byte channel chunk of a row.
    // Lane order of the move: piece = lane & 7, row = lane >> 3 -- eight consecutive lanes read one cache line.
    // (In the MFMA operand order -- row = lane & 15, piece = lane >> 4, an item = 16 rows x half a chunk -- every lane
    // quad read four different rows, four cache lines, each line was requested twice, by different instructions, and
    // the texture unit took ~47 cycles per 1-KiB wave load instead of 16: the 24 tile loads a wave of a K-split tile
    // issues at the top of every layer took 4.5k cycles to ISSUE; profiles/r04/w_*, y_*, zb_*.)
    const int sli = lane >> 3, sg = lane & 7;
    size_t srcOff[G::kItems];
    int dstOff[G::kItems];
    bool itemOk[G::kItems];
#pragma unroll
    for (int k = 0; k < G::kItems; ++k) {
        const int wid = wave + k * NWAVES;
        const int m = wid * 8 + sli;
        itemOk[k] = (wid < 2 * G::kMF) && (m < G::kRows);
        size_t grow = row0 + (itemOk[k] ? m : 0);
        if (grow > lastRow) grow = lastRow;
        srcOff[k] = grow * (size_t)A.kdim * ES + sg * 16 + (size_t)kc0 * 128;
    }
    // MX conv tiles read their input through a buffer descriptor on the workgroup's own boards: a 32-bit lane offset
    // inside the tile + a wave-uniform chunk offset, instead of a 64-bit address per item (the loop has no vector
    // registers to spare for address pairs, and no issue slots for their 64-bit adds)
    constexpr bool kTileBuf = isMx(PREC) && MODE == kConv;
    [[maybe_unused]] int srcRel[G::kItems];
    [[maybe_unused]] __amdgpu_buffer_rsrc_t xrs;
    if constexpr (kTileBuf) {
        xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(A.x) + row0 * (size_t)A.kdim * ES, 0, 0x7fffffff, 0x00027000);
#pragma unroll
        for (int k = 0; k < G::kItems; ++k) srcRel[k] = (int)(srcOff[k] - row0 * (size_t)A.kdim * ES);
    }
    auto tileLoad = [&](int k, int chunk) -> u32x4 { // item k of channel chunk `chunk` (wave-uniform)
        if constexpr (kTileBuf) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, srcRel[k], chunk * 128, COOP ? 16 /* sc1 */ : 0));
        else return *reinterpret_cast<const u32x4*>(A.x + srcOff[k] + (size_t)chunk * 128);
    };
    u32x4 st[G::kItems];
    // K-split tiles (every chunk tile of the board resident): the tile loads go out FIRST, before the
    // per-lane tables, the weight pointers and the first weight records are set up -- the prologue of these
    // small-batch kernels is a third of their run time, and it used to spend 5k cycles before the last tile
    // load was even issued (profiles/r01/h_stamps_f16m8_b64_ksplit4.txt).
    // kF16m6 conv images: the second piece of a row's lo block ([code dwords 4-5, exponent, exponent]; plane 7 of a
    // chunk buffer) is stored 8 bytes further on than its entry's slot -- pieces still tile the plane, one slot later by
    // half -- so that the 8-byte reads of the hi and the lo block's dwords 4-5 (lane groups g and g + 1 of one LDS
    // half-wave, planes 5 and 7: the same banks otherwise) fall on disjoint banks.  Staged pieces are written as two
    // 8-byte stores (8-byte aligned everywhere; the cost of one 16-byte store).  See kSplitRead in the main loop.
    // OPT-IN (make ab ABFLAGS=-DNSG_MX_SPLIT_READ), measured no faster: see kSplitRead in the main loop.
#ifdef NSG_MX_SPLIT_READ
    constexpr bool kShiftLo = (PREC == kF16m6) && MODE == kConv;
#else
    constexpr bool kShiftLo = false;
#endif
    auto stageStore = [](unsigned char* p_, const u32x4& v_) {
        if constexpr (kShiftLo) {
            *reinterpret_cast<u32x2*>(p_) = u32x2{v_.x, v_.y};
            *reinterpret_cast<u32x2*>(p_ + 8) = u32x2{v_.z, v_.w};
        } else {
            *reinterpret_cast<u32x4*>(p_) = v_;
        }
    };
    constexpr bool kResident = isMx(PREC) && KS > 1;
    constexpr int kResChunks = 8;
    // KEEP: the chunks of this workgroup's own output channels (its NWAVES / KS channel groups of 64 = two chunks each)
    constexpr int kOwnChunks = KEEP ? 2 * (NWAVES / KS) : 0;
    [[maybe_unused]] const int ownChunk0 = (int)blockIdx.y * kOwnChunks;
    u32x4 stAll[kResident ? kResChunks : 1][G::kItems];
    if constexpr (kResident) {
        // (a chunk the layer does not have -- the stem's, a 192-channel layer's last two -- is requested past the end of
        // the descriptor and comes back as zeros: one select on the lane offset instead of a branch around every load)
#pragma unroll
        for (int c = 0; c < kResChunks; ++c) {
            if (KEEP && haveOwn && c >= ownChunk0 && c < ownChunk0 + kOwnChunks) continue; // (wave-uniform: already in LDS)
#pragma unroll
            for (int k = 0; k < G::kItems; ++k)
                stAll[c][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    xrs, c < nkc ? srcRel[k] : 0x7ffffff0, c * 128, COOP ? 16 /* sc1 */ : 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        NSG_STAMP(4); // (diagnostic builds) every tile load issued
    }
#pragma unroll
    for (int k = 0; k < G::kItems; ++k) {
        const int wid = wave + k * NWAVES;
        const int c = sg;
        const int mm = itemOk[k] ? wid * 8 + sli : 0;
        // masked lanes store to their own trash slot behind both buffers (one shared slot made
        // every masked store a 64-way same-address conflict that stalled the whole LDS): the store
        // stays unconditional, so st[] stays in registers
        // (bit select, not ?: -- the compiler turned the conditional into a far out-of-line block per item)
        const int okOff = c * G::kPlaneStride + entryOfRow<G::kBoards>(mm) * 16 + ((kShiftLo && c == 7) ? 8 : 0), okMask = -(int)itemOk[k];
        dstOff[k] = (okOff & okMask) | ((G::kLds + lane * 16) & ~okMask);
    }

    // per-lane LDS read bases of the row fragments
    int abase[kMFw];
    // (edge-packed rows: eleven (fragment, lane) -> square decodes and the epilogue's row table, ~300 instructions.
    // They run AFTER the first weight and tile requests have been issued, not in front of them; the table:
    // the epilogue moves a fragment's 16 rows to and from global memory, and they are no longer consecutive
    // there -- byte offset of row r = it*4 + lrow of fragment f inside the tile at [(f*4 + lrow)*4 + it],
    // ~0 = padding row, written by the sixteen lanes that have just computed the natural row of (f, li))
    constexpr int kTabParts = KS > 1 ? KS : SS;
    constexpr int kTabFrags = (kMFw + kTabParts - 1) / kTabParts * kTabParts; // fragments the epilogue may ask the table about
    static_assert(!kPerm || (kTabFrags + 1) * 64 <= 1024, "epilogue row table");
#define NSG_COMPUTE_ABASE \
_Pragma("unroll") \
    for (int f = 0; f < kMFw; ++f) { \
        [[maybe_unused]] int eb = 0, ey = 0, ex = 0; \
        [[maybe_unused]] bool eok = true; \
        if constexpr (kPerm) eok = EdgeRows::square(f, li, eb, ey, ex); \
        const int m = kPerm ? (eok ? eb * 81 + ey * 9 + ex : -1) : (fBase + f) * 16 + li; \
        if constexpr (kPerm) { \
            static_assert(NFRAG * 16 * ES == 256, "edge-packed rows: 256-byte row slices (4 rows per instruction)"); \
            if (wave == 0 && g == 0) { \
                reinterpret_cast<unsigned*>(smem + G::kRowTabOff)[(f * 4 + (li & 3)) * 4 + (li >> 2)] = \
                    m >= 0 ? (unsigned)m * (unsigned)(A.cout * ES) : ~0u; \
                /* (fragments past the tile's last: the K or slab parts' shares of a count that does not divide) */ \
                if (f == kMFw - 1) { \
                    _Pragma("unroll") for (int fx = kMFw; fx <= kTabFrags; ++fx) \
                        reinterpret_cast<unsigned*>(smem + G::kRowTabOff)[(fx * 4 + (li & 3)) * 4 + (li >> 2)] = ~0u; \
                } \
            } \
        } \
        if constexpr (G::kBoards) { \
            const int p = kPerm ? (eok ? 24 + eb * 110 + (ey + 1) * 10 + ex : 11) \
                                : ((m < G::kRows) ? entryOfRow<true>(m) : 11); \
            abase[f] = g * G::kPlaneStride + (p - 11) * 16; \
        } else { \
            abase[f] = g * G::kPlaneStride + m * 16; \
        } \
    }
    if constexpr (!kPerm) { NSG_COMPUTE_ABASE }

    // Pins every accumulator to an AGPR at the top of a K-chunk iteration.  Without it the
    // register allocator gives the loop-carried accumulators different registers at the
    // top and the bottom of the unrolled body and rotates all 176 of them through
    // v_accvgpr_read/mov (~360 instructions per chunk, each waiting on the matrix pipe).
#define NSG_PIN_ACC_AGPR                                                                 \
    _Pragma("unroll") for (int f = 0; f < kMFw; ++f) {                                   \
        _Pragma("unroll") for (int j = 0; j < NFRAG; ++j) asm volatile("" : "+a"(acc[f][j])); \
    }
    // (macros, not lambdas: a by-reference capture keeps st[] in scratch)
#define NSG_STAGE_LOAD(KC)                                                               \
    _Pragma("unroll") for (int k = 0; k < G::kItems; ++k) {                              \
        st[k] = *reinterpret_cast<const u32x4*>(A.x + srcOff[k] + (size_t)(KC) * 128);   \
    }
#define NSG_STAGE_WRITE(BUF)                                                             \
    _Pragma("unroll") for (int k = 0; k < G::kItems; ++k) {                              \
        stageStore(smem + (itemOk[k] ? (BUF) * G::kBuf : 0) + dstOff[k], st[k]);         \
    }

    f32x4 acc[kMFw][NFRAG];
#pragma unroll
    for (int f = 0; f < kMFw; ++f)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (kM8) {
        // ---- kF16m8 main loop over PAIRS of channel chunks: per tap the triple A.m_t, B.m_t, X_t
        // (M8Seq), 27 slabs per pair, as one continuous stream of steps (slab, row fragment).
        // Row fragments live in a rolling window of kWin register slots and are requested kD
        // steps ahead (a step is 4 MFMAs: 64 clk for an f16 slab, 128 clk for an MX slab), across
        // slab and pair boundaries; the f16 weight records sit in three register sets (requested
        // two main slabs ahead), the MX records in two (X_t+1 requested at the top of X_t).  An MX
        // slab needs both chunks' tiles in LDS, so there are four image buffers: the pair being
        // computed and the pair being staged.
        using Q = M8Seq<G::kTaps>;
        using OS = OwnSeq<SS, PART>; // the slabs of a pair this wave runs, in its own order (SS = 1: all 27)
        static_assert(G::kTaps == 9, "3x3 taps");
        constexpr int kWin = 9, kD = 7; // (five or eight steps of lead: +-0.5 %, profiles/r03/README.md)
#ifdef NSG_NO_SPREAD_LOADS // A/B partner build (make ab ABFLAGS=-DNSG_NO_SPREAD_LOADS): a slab's requests behind its first MFMA
        constexpr bool kSpread = false;
#else
        constexpr bool kSpread = kMFw > NFRAG; // (fragments 0 .. NFRAG run every tap, edge-packed rows or not)
#endif
        using ST = StepSeq<OS, kMFw, kPerm, kWin, EdgeRows>;
        constexpr int kReal = ST::kReal;   // (own slab, fragment) steps that issue MFMAs
        constexpr int kSteps = ST::kSteps; // ... padded to a multiple of the window: a step's slot is the same in every pair
        constexpr int kPad = kSteps - kReal;
        static_assert(kSteps % kWin == 0 && kPad < kD, "window slot must carry across chunk pairs");
        // Next pair's tiles.  All workgroups run in lock-step, so tile loads issued at one point
        // hit HBM/MALL as one burst and take > 2 us; and VMEM loads return in order, so every
        // weight record requested behind them waits that long too.  The items are therefore
        // requested a few per slab, each AFTER its step's weight requests: chunk A' over own slabs
        // kLoadA.., written to LDS at the top of own slab kWriteA; chunk B' likewise behind it; one
        // barrier per pair, just before the first next-pair fragment request.  (All 27 slabs: A' over
        // slabs 2..7, written at 12; B' over 13..18, written at 25.)
        constexpr int kOwn = OS::kN;
        // (short own sequences: all of a chunk's items behind ONE slab's weight requests, three slabs before they are
        // written -- with a few items per slab over several slabs the last ones had a slab of lead and every pair
        // waited a tile round trip twice: 12k cycles per pair for 4.2k of MFMAs)
        constexpr int kLoadSlabs = kOwn >= 20 ? (G::kItems < 6 ? G::kItems : 6) : (kOwn >= 12 ? 3 : 1);
        constexpr int kLoadSlabA = kOwn >= 20 ? 2 : 0, kWriteA = kOwn >= 20 ? 12 : (kOwn >= 12 ? 6 : 3);
        constexpr int kLoadSlabB = kOwn >= 20 ? 13 : kWriteA, kWriteB = kOwn >= 20 ? 25 : (kOwn >= 12 ? 12 : kOwn - 1);
        static_assert(kLoadSlabA + kLoadSlabs <= kWriteA && kLoadSlabB + kLoadSlabs <= kWriteB && kWriteB < kOwn, "tile staging order");
        constexpr int kBarStep = kReal - kD;
        constexpr int kWriteStepA = ST::index(kWriteA, 0), kWriteStepB = ST::index(kWriteB, 0) < kBarStep ? ST::index(kWriteB, 0) : kBarStep - 1;
        static_assert(kWriteStepA >= 0 && ST::index(kWriteB, 0) >= 0 && kWriteStepA < kWriteStepB, "fragment 0 runs every tap");
        static_assert(kWriteStepB < kBarStep, "tile staging order");
        auto tapOff = [](int t) constexpr { return ((t / 3 - 1) * 10 + (t % 3 - 1) + 11) * 16; };
        // MX operand of lane (li, g): 32 fp8 bytes of chunk g>>1 (A / B buffer); g&1 ? lo bytes : hi bytes
        // (planes 4 + 2 * (g & 1) and the one behind it; abase holds plane g)
        const int offp8 = (g >> 1) * G::kBuf + (4 + 2 * (g & 1) - g) * G::kPlaneStride;
        // Weight records of a chunk pair, per tap: [main A: nft KiB][main B: nft KiB][MX slab].  kF16m8: the MX slab is
        // two 1-KiB records per fragment (32 fp8 bytes per lane).  kF16m6 (kPackX): per group of four fragments
        // [4 x 1 KiB: code dwords 0-3][4 x 512 B: code dwords 4-5][256 B: one dword per lane = the four fragments'
        // E8M0 exponents, byte j for fragment j (the instruction's op_sel picks it)] = 100 bytes per lane instead of
        // 128: the padding behind every 24-byte block and its exponent byte made up 11 % of a layer's weight bytes,
        // and the one-board tiles of the mid batches run at the rate their CU's L1 streams weights (1.18 MB per layer
        // and workgroup at 128 boards = 18k cycles at 64 B/clk against 20.7k cycles of MFMAs).
        constexpr bool kPackX = kM6;
        const int mainSet = nft * 1024, xSet = nft * (kPackX ? 1600 : 2048);
        const int tapStride = 2 * mainSet + xSet, pairStride = G::kTaps * tapStride;
        // (tapS / mainS: the same two numbers, made opaque at the top of every pair so that the ~40 slab offsets
        // built from them are computed where they are used -- two scalar instructions -- instead of being hoisted
        // out of the pair loop into scalar registers the loop does not have: the K-split kernels spilled them to
        // vector lanes, 120 v_readlane + their wait states per pair)
        int tapS = tapStride, mainS = mainSet;
        auto mainOff = [&](int s) { return (s / 3) * tapS + (s % 3) * mainS; }; // s % 3 in {0, 1}
        auto xOff = [&](int s) { return (s / 3) * tapS + 2 * mainS; };
        constexpr bool kStage = (KS == 1); // KS > 1: every chunk tile is resident, nothing is staged in the loop
        const int kpart = (KS > 1) ? wave % KS : 0; // this wave's share of the chunk pairs
        const int npairs = nkc / 2 / KS; // pairs this wave runs (the host pads the input channels to whole pairs)
        const int pair0 = kpart * npairs;  // K split: a contiguous range of the pairs
        // The records are read through a buffer descriptor: wave-uniform byte offset (scalar registers and the
        // instruction's immediate) + one 32-bit lane offset that never changes, instead of a 64-bit per-lane address
        // per request (a v_lshl_add_u64 each, and the address pairs live in vector registers this loop does not have).
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(A.w), 0, 0x7fffffff, 0x00027000);
        int wc = pair0 * pairStride;                                      // byte offset of the current pair's records
        const int wg4 = waveGroup * NFRAG * 1024;                         // this wave's fragments inside a main set
        const int wg8 = kPackX ? waveGroup * 6400 : waveGroup * NFRAG * 2048; // ... inside an MX slab
        const int lane16 = lane * 16, lane8 = lane * 8, lane4 = lane * 4;
        // (off_: wave-uniform, one scalar register per slab; imm_: a constant below 4096 that rides in the instruction)
        auto ld16 = [&](int off_, int imm_) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16 + imm_, off_, 0)); };
        auto ld8 = [&](int off_, int imm_) { return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(wrs, lane8 + imm_, off_, 0)); };
        auto ld4 = [&](int off_, int imm_) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane4 + imm_, off_, 0); };
        // weight registers: three sets of f16 records (own main slab o in set o % 3, requested two own main
        // slabs ahead), two sets of MX records (own MX slab j in set j & 1)
        u32x4 w4[OS::kMain > 0 ? 3 : 1][NFRAG];
        u32x4 w8[OS::kX > 0 ? 2 : 1][NFRAG][kPackX ? 1 : 2]; // code dwords 0-3 (kF16m8: both halves of the 32 bytes)
        [[maybe_unused]] u32x2 w8b[OS::kX > 0 ? 2 : 1][NFRAG];    // kPackX: code dwords 4-5
        [[maybe_unused]] uint32_t w8s[OS::kX > 0 ? 2 : 1];        // kPackX: the four fragments' exponents
        // MX records of own slab `S_` (byte offset OFF_ from wc) into set SET_: fragment J_ only, or all (J_ < 0)
#define NSG_M8_LOADX(SET_, OFF_, J_)                                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < NFRAG; ++j_) {                                                  \
            if ((J_) >= 0 && j_ != (J_)) continue;                                                              \
            if constexpr (kPackX) {                                                                             \
                w8[SET_][j_][0] = ld16(wc + (OFF_) + wg8, j_ * 1024);                                           \
                w8b[SET_][j_] = ld8(wc + (OFF_) + wg8 + 4096, j_ * 512);                                        \
                if ((J_) < 0 ? j_ == 0 : (J_) == 0) w8s[SET_] = ld4(wc + (OFF_) + wg8 + 4096, 2048);            \
            } else {                                                                                            \
                _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                \
                    w8[SET_][j_][kPackX ? 0 : h_] = ld16(wc + (OFF_) + wg8 + (j_ >> 1) * 4096, ((j_ & 1) * 2 + h_) * 1024); \
            }                                                                                                   \
        }
        if constexpr (OS::kMain > 0) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) w4[o][j] = ld16(wc + mainOff(OS::mainSlab(o)) + wg4, j * 1024);
        }
        if constexpr (OS::kXCyclic) { NSG_M8_LOADX(0, xOff(OS::xSlab(0)), -1) }

        if constexpr (kStage) {
        // first pair's tiles: both requested up front (the loop's operand registers are not live yet)
        u32x4 st1[G::kItems];
#pragma unroll
        for (int k = 0; k < G::kItems; ++k) st[k] = tileLoad(k, 0);
#pragma unroll
        for (int k = 0; k < G::kItems; ++k) st1[k] = tileLoad(k, 1);
        NSG_STAMP(4);
        if constexpr (kPerm) { NSG_COMPUTE_ABASE }
        if (zeroLds) zeroHalo<G, kShiftLo>(smem, tid);
        __syncthreads(); // zero fill done before staging writes
        NSG_STAMP(5);
        NSG_STAGE_WRITE(0)
#pragma unroll
        for (int k = 0; k < G::kItems; ++k)
            stageStore(smem + (itemOk[k] ? G::kBuf : 0) + dstOff[k], st1[k]);
        __syncthreads();
        } else {
            // every chunk of the board (at most eight) was requested at the top of the kernel
            constexpr int kChunks = kResChunks;
            if constexpr (kPerm) { NSG_COMPUTE_ABASE }
            if (zeroLds) zeroHalo<G, kShiftLo>(smem, tid);
            __syncthreads();
            NSG_STAMP(5); // image cleared; what follows waits for the tile loads
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                if (KEEP && haveOwn && c >= ownChunk0 && c < ownChunk0 + kOwnChunks) continue; // (the epilogue before this layer left them here)
#pragma unroll
                for (int k = 0; k < G::kItems; ++k)
                    if (__builtin_expect(c < nkc, 1)) stageStore(smem + (itemOk[k] ? c * G::kBuf : 0) + dstOff[k], stAll[c][k]);
            }
            __syncthreads();
        }
        NSG_STAMP(1);

        // The MX row operand of the e2m3 form is six code dwords + the block's exponent (which must sit in an arch
        // VGPR).  Read as two 16-byte pieces, the second is half codes, half exponent: the register allocator, out of
        // arch VGPRs, loads the codes elsewhere and COPIES them next to the first half -- 138 moves + 87 s_nop per chunk
        // pair in front of the MX steps' first MFMAs.  kSplitRead (-DNSG_MX_SPLIT_READ) reads 16 bytes (dwords 0-3) + 8
        // bytes (dwords 4-5), both wholly inside the operand, + 4 bytes (the exponent's second copy, at byte 12 of the
        // piece: a read of byte 8 would be fused with the 8-byte read into a 12-byte one and copied again), with the lo
        // block's second piece 8 bytes off its slot (kShiftLo) so that the narrow reads of lane groups g, g + 1 use
        // disjoint banks.  The loop then has no move, no s_nop and no accumulator-file copy left -- and is NOT faster:
        // 512 boards -1.3 %, 32-64 boards -1.6...-3 %, the rest +-0.5 % (profiles/r04/m_ab_shifted_plane_split_mx_read.txt;
        // without the shift -3...-4 %, d_ab_*).  The copies issue in the shadow of the step's MFMAs; a third LDS
        // instruction per MX step does not.  What the loop is short of is LDS issue, not vector issue.
        constexpr bool kSplitRead = kShiftLo && OS::kX > 0;
        // (without the split read, plane 7's shift still applies: its piece is read at + 8)
        const int offB = offp8 + G::kPlaneStride + 8 * (g & 1);
        u32x4 aw[kWin][kSplitRead ? 1 : 2];
        [[maybe_unused]] u32x2 aw45[kSplitRead ? kWin : 1];
        [[maybe_unused]] uint32_t awS[kSplitRead ? kWin : 1];
        // fragment request of step q (q >= kSteps: the next pair's step q - kSteps; null steps request nothing)
#define NSG_M8_REQ(QQ, CUR, NXT)                                                                  \
        if (ST::frag((QQ) % kSteps) >= 0) {                                                       \
            const int q_ = (QQ) % kSteps;                                                         \
            const unsigned char* b_ = ((QQ) >= kSteps) ? (NXT) : (CUR);                           \
            const int s_ = OS::slab(ST::pos(q_)), f_ = ST::frag(q_);                              \
            if (Q::isX(s_)) {                                                                     \
                const unsigned char* ap_ = b_ + abase[f_] + offp8 + tapOff(Q::tap(s_));           \
                aw[(QQ) % kWin][0] = *reinterpret_cast<const u32x4*>(ap_);                        \
                [[maybe_unused]] const unsigned char* ap2_ = b_ + abase[f_] + offB + tapOff(Q::tap(s_)); \
                if constexpr (kSplitRead) {                                                       \
                    aw45[(QQ) % kWin] = *reinterpret_cast<const u32x2*>(ap2_);                    \
                    awS[(QQ) % kWin] = *reinterpret_cast<const uint32_t*>(ap2_ + 12);             \
                } else if constexpr (kShiftLo) {                                                  \
                    aw[(QQ) % kWin][kSplitRead ? 0 : 1] = *reinterpret_cast<const u32x4*>(ap2_);  \
                } else {                                                                          \
                    aw[(QQ) % kWin][kSplitRead ? 0 : 1] = *reinterpret_cast<const u32x4*>(ap_ + G::kPlaneStride); \
                }                                                                                 \
            } else {                                                                              \
                aw[(QQ) % kWin][0] = *reinterpret_cast<const u32x4*>(                             \
                    b_ + Q::half(s_) * G::kBuf + abase[f_] + tapOff(Q::tap(s_)));                 \
            }                                                                                     \
        }
        const unsigned char* first = kStage ? smem : smem + (size_t)(2 * pair0) * G::kBuf;
        // Requests run kD steps ahead.  A step whose request would fall on one of the kPad null steps at the
        // end of the previous pair is requested at the top of its own pair instead (step 0 requests steps
        // kD - kPad .. kD): the fill before the loop leaves those to it.
#pragma unroll
        for (int q = 0; q < kD - kPad; ++q) NSG_M8_REQ(q, first, first)

        for (int kp = 0; kp < npairs; ++kp) {
            const unsigned char* abuf = kStage ? smem + ((2 * kp) & 3) * G::kBuf : smem + (size_t)(2 * (pair0 + kp)) * G::kBuf;
            // (K split, last pair: the look-ahead requests read this pair again, harmlessly)
            const unsigned char* nbuf = kStage ? smem + ((2 * kp + 2) & 3) * G::kBuf
                                               : (kp + 1 < npairs ? abuf + 2 * G::kBuf : abuf);
            [[maybe_unused]] const int nextA = (kp + 1 < npairs) ? 2 * kp + 2 : 2 * kp; // (last pair: harmless re-load)
            // records of the NEXT pair (the last pair re-reads its own: the buffer's zero padding covers two
            // record sets past the end, not a whole pair)
            [[maybe_unused]] const int nextRec = (SS == 1 || kp + 1 < npairs) ? pairStride : 0;
            asm volatile("" : "+s"(tapS), "+s"(mainS));
            NSG_PIN_ACC_AGPR
#pragma unroll
            for (int u = 0; u < kOwn; ++u) {
#pragma unroll
                for (int f = 0; f < kMFw; ++f) {
                    if (!ST::active(u, f)) continue; // (edge-packed rows: this fragment has no neighbours under this tap)
                    const int s = OS::slab(u);
                    const int q = ST::index(u, f);
#ifdef NSG_DIAG_SLABS
                    // slab timeline of workgroup 0 / wave 0, stored behind the per-workgroup stamps
                    // (its s_memtime drains the fragment window: a separate diagnostic build)
                    if (f == 0 && A.stamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
                        A.stamps[2048 + kp * 32 + s] = __builtin_amdgcn_s_memtime();
#endif
                    if (kStage && q == kWriteStepA) { NSG_STAGE_WRITE((2 * kp + 2) & 3) }
                    if (kStage && q == kWriteStepB) { NSG_STAGE_WRITE((2 * kp + 3) & 3) }
                    if (kStage && q == kBarStep) __syncthreads(); // publishes the two tiles written above
                    // MX records, two sets: cyclic (the next own MX slab, the next pair's first behind the last, is
                    // requested at the top of the current one), or the pair's first at the top of the pair (set 0 is
                    // free once the previous pair's last MX slab is done) and X_j+1 at the top of X_j.
                    const int xj = OS::xOrd(u);
                    const bool reqX = OS::kXCyclic ? xj >= 0 : ((u == 0 && OS::kX > 0) || (xj >= 0 && xj + 1 < OS::kX));
                    // Record and tile requests of a slab.  kSpread: one fragment's records per step over the slab's first
                    // NFRAG steps (a vector-memory instruction occupies the wave's issue port longer than the half of an
                    // MFMA's cycles that are free: four to eight of them behind ONE MFMA hold the next MFMA back; one
                    // or two per step hide behind that step's own MFMAs) and the tile items in the step behind them.
                    // (the step that carries a slab's tile requests must exist in every slab: with one board's edge-packed
                    // rows only fragments 0 .. 3 run every tap -- the requests share fragment 3's step with its records)
                    const int fTile = kSpread ? (kPerm && SIZE == 1 ? NFRAG - 1 : NFRAG) : 0;
                    static_assert(!kPerm || SIZE != 1 || (ST::active(0, NFRAG - 1) && ST::active(8, NFRAG - 1)), "tile-request step");
                    if (kSpread ? f < NFRAG : f == 0) {
                        if (!Q::isX(s)) { // f16 record two own main slabs ahead (beyond this pair: the next pair's)
                            const int o2 = OS::mainOrd(u) + 2;
                            const int o = (o2 >= OS::kMain ? nextRec : 0) + mainOff(OS::mainSlab(o2 % (OS::kMain > 0 ? OS::kMain : 1)));
#pragma unroll
                            for (int j = 0; j < NFRAG; ++j)
                                if (!kSpread || j == f) w4[o2 % 3][j] = ld16(wc + o + wg4, j * 1024);
                        }
                        if (reqX) {
                            const int j2 = OS::kXCyclic ? xj + 1 : ((u == 0 && xj != 0) ? 0 : xj + 1);
                            const int o = ((OS::kXCyclic && j2 >= OS::kX) ? nextRec : 0) + xOff(OS::xSlab(j2 % (OS::kX > 0 ? OS::kX : 1)));
                            NSG_M8_LOADX(j2 & 1, o, (kSpread ? f : -1))
                        }
                    }
                    const bool loadA = kStage && f == fTile && u >= kLoadSlabA && u < kLoadSlabA + kLoadSlabs;
                    const bool loadB = kStage && f == fTile && u >= kLoadSlabB && u < kLoadSlabB + kLoadSlabs;
                    if (loadA || loadB) {
#pragma unroll
                        for (int k = u - (loadA ? kLoadSlabA : kLoadSlabB); k < G::kItems; k += kLoadSlabs)
#ifdef NSG_EXP_RUNTIME // bit 2 (timing only, wrong results): every in-loop tile item re-reads one hot line
                            if (A.exp & 4) st[k] = *reinterpret_cast<const u32x4*>(A.x + lane * 16);
                            else
#endif
                            st[k] = tileLoad(k, nextA + (loadB ? 1 : 0));
                    }
                    if (q == 0) {
#pragma unroll
                        for (int r = kD - kPad; r < kD; ++r) NSG_M8_REQ(r, abuf, nbuf)
                    }
                    NSG_M8_REQ(q + kD, abuf, nbuf)
                    const int slot = q % kWin;
                    if (Q::isX(s)) {
                        // kSplitRead: the operand's six data dwords arrive as a 16-byte and an 8-byte read -- both
                        // wholly inside the six-register operand, so they land in place -- and the block's exponent
                        // as a third, 4-byte read of the block's LAST dword (the exponent is stored twice, at bytes
                        // 24 and 28: a read of byte 24 is adjacent to the 8-byte one and the compiler fuses the two
                        // into one 12-byte read again).  Two 16-byte reads leave the second one half data, half
                        // exponent: under register pressure its data half was copied into place -- two moves and
                        // the wait states between a vector write and the MFMA that reads it, in front of EVERY MX
                        // step's first MFMA (88 moves + 80 s_nop per chunk pair, gone; profiles/r04/README.md).
                        i32x8 xb;
                        if constexpr (kSplitRead)
                            xb = i32x8{(int)aw[slot][0].x, (int)aw[slot][0].y, (int)aw[slot][0].z, (int)aw[slot][0].w,
                                       (int)aw45[slot].x, (int)aw45[slot].y, (int)awS[slot], 0};
                        else xb = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, aw[slot][0]),
                                                          __builtin_bit_cast(i32x4_t, aw[slot][1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j) {
                            if constexpr (kPackX) {
                                // e2m3 x e2m3: the weight block's E8M0 exponent is byte j of the slab's exponent dword
                                // (op_sel: an immediate), the row block's is byte 0 of dword 6 of its 32 bytes
                                const u32x4 a03 = w8[xj & 1][j][0];
                                const u32x2 a45 = w8b[xj & 1][j];
                                const i32x8 wa = {(int)a03.x, (int)a03.y, (int)a03.z, (int)a03.w, (int)a45.x, (int)a45.y, 0, 0};
                                const int sc = (int)w8s[xj & 1];
                                if (j == 0) acc[f][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb, acc[f][j], 2, 2, 0, sc, 0, xb[6]);
                                else if (j == 1) acc[f][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb, acc[f][j], 2, 2, 1, sc, 0, xb[6]);
                                else if (j == 2) acc[f][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb, acc[f][j], 2, 2, 2, sc, 0, xb[6]);
                                else acc[f][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb, acc[f][j], 2, 2, 3, sc, 0, xb[6]);
                            } else {
                                const i32x8 wa = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, w8[xj & 1][j][0]),
                                                                         __builtin_bit_cast(i32x4_t, w8[xj & 1][j][kPackX ? 0 : 1]), 0, 1, 2, 3, 4, 5, 6, 7);
                                acc[f][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb, acc[f][j], 0, 0, 0,
                                                                                            kM8ScaleByte, 0, 127);
                            }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j)
                            acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                __builtin_bit_cast(f16x8, w4[OS::mainOrd(u) % 3][j]), __builtin_bit_cast(f16x8, aw[slot][0]), acc[f][j], 0, 0, 0);
                    }
                    // issue order inside the step: the first MFMA, then the requests (they issue in its
                    // shadow instead of between two steps), then the other MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ST::frag((q + kD) % kSteps) >= 0) {
                        if (Q::isX(OS::slab(ST::pos((q + kD) % kSteps)))) __builtin_amdgcn_sched_group_barrier(0x100, kSplitRead ? 3 : 2, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if ((kSpread ? f < NFRAG : f == 0) && !Q::isX(s)) __builtin_amdgcn_sched_group_barrier(0x020, kSpread ? 1 : NFRAG, 0);
                    if ((kSpread ? f < NFRAG : f == 0) && reqX) {
                        if (kPackX && f == 0) __builtin_amdgcn_sched_group_barrier(0x020, (kSpread ? 2 : 2 * NFRAG) + 1, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x020, kSpread ? 2 : 2 * NFRAG, 0);
                    }
                    if (loadA || loadB) __builtin_amdgcn_sched_group_barrier(0x020, (G::kItems + kLoadSlabs - 1) / kLoadSlabs, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NFRAG - 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wc += pairStride;
        }
#undef NSG_M8_REQ
#undef NSG_M8_LOADX
        __syncthreads(); // every wave is done reading before the epilogue reuses LDS
        if constexpr (KS > 1) {
            // Sum the K parts: every wave publishes its accumulators, then keeps the row fragments
            // kpart*kMFe .. +kMFe-1 of its channel group, added up in part order (deterministic).
            // (KS = 4 on six fragments: two per wave, the last wave's two past the tile -- it adds up
            // whatever lies there and its stores are masked like any padded row)
            constexpr int kMFe = (kMFw + KS - 1) / KS;
            u32x4* xb = reinterpret_cast<u32x4*>(smem);
#pragma unroll
            for (int f = 0; f < kMFw; ++f)
#pragma unroll
                for (int j = 0; j < NFRAG; ++j)
                    xb[((wave * kMFw + f) * NFRAG + j) * 64 + lane] = __builtin_bit_cast(u32x4, acc[f][j]);
            __syncthreads();
            const int w0 = wave - kpart; // first wave of this channel group
#pragma unroll
            for (int f = 0; f < kMFe; ++f)
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    f32x4 sum = __builtin_bit_cast(f32x4, xb[(((w0 + 0) * kMFw + kpart * kMFe + f) * NFRAG + j) * 64 + lane]);
#pragma unroll
                    for (int p2 = 1; p2 < KS; ++p2)
                        sum += __builtin_bit_cast(f32x4, xb[(((w0 + p2) * kMFw + kpart * kMFe + f) * NFRAG + j) * 64 + lane]);
                    acc[f][j] = sum;
                }
            __syncthreads(); // the staging regions of the epilogue overlap the exchange area
        }
        if constexpr (SS > 1) {
            // Sum the slab parts: the SS waves of a channel group publish their accumulators in LDS and
            // wave `PART` keeps the row fragments PART*kMFe .. +kMFe-1, added up in part order
            // (deterministic).  Eleven fragments of four waves are 176 KiB: two rounds of six fragments.
            constexpr int kMFe = (kMFw + SS - 1) / SS, kFR = 6;
            static_assert(kFR % kMFe == 0 && 2 * kFR >= kMFw && NWAVES * kFR * NFRAG * 1024 <= G::kLds, "exchange rounds");
            u32x4* xb = reinterpret_cast<u32x4*>(smem);
            const int w0 = wave - PART; // first wave of this channel group
            f32x4 mine[kMFe][NFRAG];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int f = r * kFR; f < (r + 1) * kFR && f < kMFw; ++f)
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j)
                        xb[((wave * kFR + (f - r * kFR)) * NFRAG + j) * 64 + lane] = __builtin_bit_cast(u32x4, acc[f][j]);
                __syncthreads();
                if (PART * kMFe >= r * kFR && PART * kMFe < (r + 1) * kFR) {
#pragma unroll
                    for (int fe = 0; fe < kMFe; ++fe)
#pragma unroll
                        for (int j = 0; j < NFRAG; ++j) {
                            const int gf = PART * kMFe + fe; // (past the tile: zeros, its rows are masked anyway)
                            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (gf < kMFw) {
#pragma unroll
                                for (int p2 = 0; p2 < SS; ++p2)
                                    sum += __builtin_bit_cast(f32x4, xb[(((w0 + p2) * kFR + (gf - r * kFR)) * NFRAG + j) * 64 + lane]);
                            }
                            mine[fe][j] = sum;
                        }
                }
                __syncthreads(); // the next round (or the epilogue's staging regions) overwrites the exchange area
            }
#pragma unroll
            for (int fe = 0; fe < kMFe; ++fe)
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) acc[fe][j] = mine[fe][j];
        }
    } else {
    // weight stream: record q = (kc*taps + tap)*2 + s, NFRAG 1-KiB records per wave, held
    // in a register ring (a record is requested whole slabs of MFMAs before its first use,
    // which is what covers the L2 latency).  16-bit / f32: one record per slab.  kF16x3: two
    // records per tap (w_hi, w_lo) for three slabs -- w_hi stays in registers for its
    // second product instead of being streamed twice.
    constexpr int kSlabsPerTap = kSplit ? 3 : 2;
    constexpr int kSlabs = kSlabsPerTap * G::kTaps;    // slabs per channel chunk
    // LDS byte offset of slab s's row fragments inside the chunk image: tap shift +
    // which 4 of the 8 16-byte pieces (kF16x3: pieces 0-3 = hi, 4-7 = lo; slabs
    // (w_hi,x_hi) (w_lo,x_hi) (w_hi,x_lo) -> hi, hi, lo)
    auto slabOff = [](int s) constexpr {
        const int t = s / kSlabsPerTap;
        const int r = s % kSlabsPerTap;
        const int piece = kSplit ? (r == 2 ? 4 : 0) : r * 4;
        return (G::kBoards ? ((t / 3 - 1) * 10 + (t % 3 - 1) + 11) * 16 : 0) + piece * G::kPlaneStride;
    };
    // Ring depths: a slab of a full tile is 44 MFMAs (~700 cycles) and covers an L2
    // round trip with one slab of lead; a small-batch tile has as few as 6 MFMAs per
    // slab, so its weight ring runs 8 slabs ahead and its row fragments 2 slabs ahead.
    constexpr bool kDeep = (NFRAG <= 2) && (kSlabs % 9 == 0);
    constexpr int kRing = kDeep ? 9 : ((kSlabs % 3 == 0) ? 3 : 2);   // weight records (non-split)
    constexpr int kRingA = (kDeep || (NFRAG <= 2 && kSlabs % 3 == 0)) ? 3 : 2; // row fragments
    constexpr int kMfmaPerPair = (PREC == kFp32) ? 4 : 1;
    // kF16x3 weight registers.  kWMode 1 (full tiles): three sets -- w_lo in set 1, w_hi
    // alternating between sets 0 and 2 tap by tap; hi_t+1 is requested during slab (t,0)
    // and lo_t+1 during (t,2) (three and two slabs of lead).  With an odd number of taps
    // per chunk the next chunk's hi_0 lands in set 2 and is moved to set 0 at the chunk
    // boundary.  kWMode 2 (small tiles): kWR records, record u in set u % kWR, requested
    // kWR/2 - 1 taps ahead (a one-fragment wave has 6 MFMAs per slab to hide a load behind).
    constexpr int kWMode = !kSplit ? 0 : (kDeep ? 2 : 1);
    constexpr int kWR = (NFRAG == 1) ? 18 : 6;
    constexpr int kWSets = kWMode == 0 ? kRing : (kWMode == 1 ? 3 : kWR);
    const size_t slabStride = (size_t)nft * 64;
    const u32x4* wp = A.w + (size_t)waveGroup * NFRAG * 64 + lane + (size_t)kc0 * (2 * G::kTaps) * slabStride;
    u32x4 w[kWSets][NFRAG];
    constexpr int kLead = kWMode == 0 ? ((kRing == 2) ? 2 : kRing - 1) : (kWMode == 1 ? 2 : kWR - 2); // records requested up front
#pragma unroll
    for (int q = 0; q < kLead; ++q)
#pragma unroll
        for (int j = 0; j < NFRAG; ++j) w[q % kWSets][j] = wp[(size_t)q * slabStride + j * 64];
    wp += (size_t)kLead * slabStride;

    NSG_STAGE_LOAD(0)
    if constexpr (G::kBoards) {
        // zero the halo of the LDS image (it stays zero for the whole kernel) while the
        // first tile and the first weight slabs are in flight
        if (zeroLds) zeroHalo<G>(smem, tid);
    }
    if constexpr (G::kBoards) __syncthreads(); // zero fill done before staging writes
    NSG_STAGE_WRITE(0)
    __syncthreads();

    NSG_STAMP(1);
    u32x4 a[kRingA][kMFw]; // row fragments: current slab + the next kRingA-1
    // Long chunks (3x3 taps) run one continuous slab pipeline across chunk boundaries:
    // the next chunk's tile is written to the other LDS buffer a third of the way into
    // the chunk, the barrier that publishes it sits kRingA slabs before the end, and the
    // last slabs already request the next chunk's first row fragments -- no drain, no
    // LDS write latency and no cold fragment read between chunks.
    constexpr bool kFlow = (kSlabs >= 9);
    // kF16x3: slabs (w_hi,x_hi) and (w_lo,x_hi) of a tap share their row fragments, so a
    // tap needs two fragment sets (hi, lo), not three.  Set u = 2*tap + (lo ? 1 : 0)
    // lives in a[u % kSlotR]:
    //   two sets (full tiles):  lo_t is requested during slab (t,0), hi_t+1 during (t,2);
    //   three sets (small tiles): hi_t+1 during (t,0), lo_t+1 during (t,2) -- three slabs
    //   of lead, which a slab of only 6 or 12 MFMAs needs to cover the LDS latency.
    constexpr int kSlotR = (kFlow && kSplit) ? kRingA : 0;
    constexpr bool kSlot = kSlotR > 0;
    static_assert(!kFlow || kSlot || kSlabs % kRingA == 0, "ring slot must carry across chunks");
    static_assert(!kSlot || (2 * G::kTaps) % kSlotR == 0, "fragment sets must carry across chunks");
    // barrier at the top of the first slab that reads the next chunk's buffer
    constexpr int kBarSlab = kSlot ? (kSlotR == 2 ? kSlabs - 1 : kSlabs - 3) : kSlabs - kRingA + 1;
    // Next chunk's tile: its items are requested one per slab over the first slabs, each
    // BEHIND that slab's weight request (VMEM returns in order: a weight record queued
    // behind the lock-step tile burst waits an HBM round trip with it), and written to
    // LDS in the slab before the barrier.
    constexpr int kLoadSlabs = G::kItems < 6 ? G::kItems : 6;
    constexpr int kItemsPerSlab = (G::kItems + kLoadSlabs - 1) / kLoadSlabs;
    constexpr int kWriteSlab = kBarSlab - 1;
    static_assert(!kFlow || kLoadSlabs + 2 < kWriteSlab, "tile staging order");
    if constexpr (kFlow) {
#pragma unroll
        for (int q = 0; q < (kSlot ? kSlotR - 1 : kRingA - 1); ++q)
#pragma unroll
            for (int f = 0; f < kMFw; ++f)
                a[q][f] = *reinterpret_cast<const u32x4*>(smem + abase[f] + slabOff(kSlot ? 2 * q : q));
    }
    for (int kc = 0; kc < nkc; ++kc) {
        const unsigned char* abuf = smem + (kc & 1) * G::kBuf;
        const unsigned char* nbuf = smem + ((kc + 1) & 1) * G::kBuf;
        if constexpr (!kFlow) {
#pragma unroll
            for (int q = 0; q < kRingA - 1; ++q) // first slabs of this chunk (just published by the barrier)
#pragma unroll
                for (int f = 0; f < kMFw; ++f)
                    if (q < kSlabs) a[q][f] = *reinterpret_cast<const u32x4*>(abuf + abase[f] + slabOff(q));
        }
        // next chunk's tile: global -> registers, registers -> LDS later
        // (the last iteration re-loads its own chunk: harmless, keeps st[] in registers)
        if constexpr (!kFlow) { NSG_STAGE_LOAD(kc + 1 < nkc ? kc + 1 : kc) }
#pragma unroll
        for (int s = 0; s < kSlabs; ++s) {
            if constexpr (kFlow) {
                if (s == kWriteSlab) { NSG_STAGE_WRITE((kc + 1) & 1) }
                if (s == kBarSlab) __syncthreads();
            }
            // -- requests for later slabs
            bool dsReads = true;
            if constexpr (kSlot) {
                const int r = s % 3, t = s / 3;
                dsReads = (r != 1);
                // which set to request: (tap, lo?) and where it goes
                const int nt = (kSlotR == 2) ? (r == 0 ? t : t + 1) : t + 1;
                const bool nlo = (kSlotR == 2) ? (r == 0) : (r == 2);
                if (r != 1) {
                    const unsigned char* src = (nt < G::kTaps) ? abuf : nbuf;
                    const int off1 = slabOff(3 * (nt % G::kTaps) + (nlo ? 2 : 0));
#pragma unroll
                    for (int f = 0; f < kMFw; ++f)
                        a[(2 * nt + (nlo ? 1 : 0)) % kSlotR][f] =
                            *reinterpret_cast<const u32x4*>(src + abase[f] + off1);
                }
            } else if (s + kRingA - 1 < kSlabs) {
                const int off1 = slabOff(s + kRingA - 1);
#pragma unroll
                for (int f = 0; f < kMFw; ++f)
                    a[(s + kRingA - 1) % kRingA][f] = *reinterpret_cast<const u32x4*>(abuf + abase[f] + off1);
            } else if constexpr (kFlow) {
                // first slabs of the next chunk (after the last chunk: a harmless re-read)
                const int off1 = slabOff(s + kRingA - 1 - kSlabs);
#pragma unroll
                for (int f = 0; f < kMFw; ++f)
                    a[(s + kRingA - 1) % kRingA][f] = *reinterpret_cast<const u32x4*>(nbuf + abase[f] + off1);
            }
            // -- weight records for later slabs, and which set this slab multiplies by
            int wset = s % kRing;
            bool wLoads = true;
            if constexpr (kWMode == 0) {
                if constexpr (kRing >= 3) {
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) w[(s + kRing - 1) % kRing][j] = wp[j * 64];
                }
            } else {
                const int r = s % 3, t = s / 3;
                int dst = -1;
                if constexpr (kWMode == 1) {
                    wset = (r == 1) ? 1 : (t & 1) * 2;
                    if (r == 0) dst = ((t + 1) & 1) * 2; // hi of the next tap
                    if (r == 2) dst = 1;                 // lo of the next tap
                } else {
                    wset = (2 * t + (r == 1 ? 1 : 0)) % kWR;
                    if (r == 0) dst = (2 * t + kWR - 2) % kWR;   // hi, kWR/2 - 1 taps ahead
                    if (r == 1) dst = (2 * t + kWR - 1) % kWR;   // lo
                }
                wLoads = dst >= 0;
                if (dst >= 0) {
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) w[dst][j] = wp[j * 64];
                    wp += slabStride;
                }
            }
            if constexpr (kFlow) {
                if (s < kLoadSlabs) {
#pragma unroll
                    for (int k = s; k < G::kItems; k += kLoadSlabs)
                        st[k] = *reinterpret_cast<const u32x4*>(A.x + srcOff[k] + (size_t)(kc + 1 < nkc ? kc + 1 : kc) * 128);
                }
            }
            // -- this slab's MFMAs
            const int aslot = kSlot ? (2 * (s / 3) + (s % 3 == 2 ? 1 : 0)) % kSlotR : s % kRingA;
#pragma unroll
            for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                for (int f = 0; f < kMFw; ++f) mfmaSlab<PREC>(acc[f][j], w[wset][j], a[aslot][f]);
            if constexpr (kWMode == 0) {
                if constexpr (kRing == 2) {
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) w[s % 2][j] = wp[j * 64];
                }
                wp += slabStride;
            }
            if constexpr (kRing >= 3) {
                // interleave: one LDS read (+ one weight load) per NFRAG row-fragment MFMAs
#pragma unroll
                for (int f = 0; f < kMFw; ++f) {
                    if (dsReads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // DS read
                    if (wLoads && f < NFRAG) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // VMEM read
                    if (kFlow && s < kLoadSlabs && f == NFRAG) __builtin_amdgcn_sched_group_barrier(0x020, kItemsPerSlab, 0); // tile items
                    if (kFlow && s == kWriteSlab && f < G::kItems)
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); // DS write
                    __builtin_amdgcn_sched_group_barrier(0x008, NFRAG * kMfmaPerPair, 0); // MFMA
                }
            }
            __builtin_amdgcn_sched_barrier(0); // nothing sinks out of its slab
        }
        if constexpr (kWMode == 1 && (G::kTaps & 1)) {
#pragma unroll
            for (int j = 0; j < NFRAG; ++j) w[0][j] = w[2][j]; // next chunk's hi_0
        }
        if constexpr (!kFlow) {
            NSG_STAGE_WRITE((kc + 1) & 1)
            __syncthreads();
        }
    }
    if constexpr (kFlow) __syncthreads(); // every wave is done reading before the epilogue reuses LDS
    }

#undef NSG_STAGE_LOAD
#undef NSG_COMPUTE_ABASE
#undef NSG_PIN_ACC_AGPR
#undef NSG_STAGE_WRITE
    NSG_STAMP(2);

    // ---- epilogue.  Lane (li, g) holds, for fragment j, channels
    // cbase + 4j + {0..3} of row f*16 + li.
    // (weights are packed in groups of kNfrag = 4 fragments = 64 channels: fragment
    // nf, MFMA row 4g+r  <->  channel (nf/4)*64 + g*16 + (nf%4)*4 + r; a wave that runs
    // NFRAG < 4 fragments owns 4*NFRAG consecutive channels inside such a group)
    const int nf0 = waveGroup * NFRAG;
    const int cbase = (nf0 >> 2) * 64 + g * 16 + (nf0 & 3) * 4;
    float bv[NFRAG * 4];
#pragma unroll
    for (int i = 0; i < NFRAG * 4; ++i) bv[i] = A.bias[cbase + i];

    if constexpr (MODE == kConv && NFRAG < 4) {
        // ---- small-batch tiles (1 or 2 fragments per wave): direct stores from the
        // MFMA layout; these launches are latency-bound, not bandwidth-bound.
#pragma unroll
        for (int f = 0; f < kMFw; ++f) {
            const int m = (fBase + f) * 16 + li;
            if (m >= G::kRows) continue;
            const size_t grow = row0 + m;
            float v[NFRAG * 4];
#pragma unroll
            for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j * 4 + r] = fmaf(acc[f][j][r], A.accScale, bv[j * 4 + r]);
            if constexpr (kSplit) {
                const size_t e = grow * (size_t)A.cout * 4 + (size_t)(cbase >> 5) * 128 + (cbase & 31) * 2;
                if (hasRes) {
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) {
                        const u32x2 rh = *reinterpret_cast<const u32x2*>(A.res + e + j * 8);
                        const u32x2 rl = *reinterpret_cast<const u32x2*>(A.res + e + 64 + j * 8);
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            v[j * 4 + 2 * i] += f16BitsToF32((uint16_t)(rh[i] & 0xffffu)) + f16BitsToF32((uint16_t)(rl[i] & 0xffffu));
                            v[j * 4 + 2 * i + 1] += f16BitsToF32((uint16_t)(rh[i] >> 16)) + f16BitsToF32((uint16_t)(rl[i] >> 16));
                        }
                    }
                }
                if (A.relu) {
#pragma unroll
                    for (int i = 0; i < NFRAG * 4; ++i) v[i] = reluF(v[i]);
                }
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    u32x2 oh, ol;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        uint16_t h0, l0, h1, l1;
                        splitF16(v[j * 4 + 2 * i], h0, l0);
                        splitF16(v[j * 4 + 2 * i + 1], h1, l1);
                        oh[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
                        ol[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
                    }
                    *reinterpret_cast<u32x2*>(A.y + e + j * 8) = oh;
                    *reinterpret_cast<u32x2*>(A.y + e + 64 + j * 8) = ol;
                }
            } else {
                const size_t e = (grow * (size_t)A.cout + cbase) * ES;
                if (hasRes) {
#pragma unroll
                    for (int j = 0; j < NFRAG; ++j) {
                        if constexpr (PREC == kFp32) {
                            const f32x4 rr = *reinterpret_cast<const f32x4*>(A.res + e + j * 16);
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[j * 4 + i] += rr[i];
                        } else {
                            const u32x2 rr = *reinterpret_cast<const u32x2*>(A.res + e + j * 8);
                            v[j * 4 + 0] += unpackLo<PREC>(rr.x); v[j * 4 + 1] += unpackHi<PREC>(rr.x);
                            v[j * 4 + 2] += unpackLo<PREC>(rr.y); v[j * 4 + 3] += unpackHi<PREC>(rr.y);
                        }
                    }
                }
                if (A.relu) {
#pragma unroll
                    for (int i = 0; i < NFRAG * 4; ++i) v[i] = reluF(v[i]);
                }
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) {
                    if constexpr (PREC == kFp32) {
                        *reinterpret_cast<f32x4*>(A.y + e + j * 16) = f32x4{v[j * 4], v[j * 4 + 1], v[j * 4 + 2], v[j * 4 + 3]};
                    } else {
                        *reinterpret_cast<u32x2*>(A.y + e + j * 8) =
                            u32x2{packPair<PREC>(v[j * 4], v[j * 4 + 1]), packPair<PREC>(v[j * 4 + 2], v[j * 4 + 3])};
                    }
                }
            }
        }
        NSG_STAMP(3);
        return;
    }

#ifdef NSG_EXP_RUNTIME // timing-only (wrong results), switched at run time so that the activations the
    // kernels read stay those of the last correct forward: bit 1 = skip the whole epilogue
    if constexpr (MODE == kConv && NFRAG == 4 && isMx(PREC)) {
        if (A.exp & 2) {
#pragma unroll
            for (int f = 0; f < kMFw; ++f)
#pragma unroll
                for (int j = 0; j < NFRAG; ++j) asm volatile("" ::"v"(acc[f][j]));
            NSG_STAMP(3);
            return;
        }
    }
#endif
    if constexpr (MODE == kConv && NFRAG == 4) {
        // (K split: after the exchange a wave owns kMFe of the row fragments, from fBaseE on)
        constexpr int kParts = KS > 1 ? KS : SS;
        constexpr int kMFe = (kMFw + kParts - 1) / kParts;
        const int fBaseE = fBase + ((KS > 1) ? (wave % KS) * kMFe : PART * kMFe);
        // ---- convolution epilogue, staged through LDS so that every global access is
        // a full-line, lane-linear 16-byte access.  In the MFMA result layout a lane
        // owns 16 channels of ONE row, so a direct store scatters 16-byte pieces over
        // 16 rows per instruction (measured: ~25 us of a 110 us layer).  Instead each
        // wave builds the memory image of its 64-channel slice of 16 rows per fragment
        // in a private LDS region (the input buffers are free after the last chunk's
        // barrier), then moves it with lane l <-> piece l of the row-major image.
        constexpr int kRowB = NFRAG * 16 * ES;      // this wave's bytes per row: 256 or 128
        constexpr int kRowS = kRowB + 16;           // padded LDS row stride
        constexpr int kNP = kRowB / 64;             // 16-byte pieces per lane: 4 or 2
        constexpr int kPPR = kRowB / 16;            // pieces per row: 16 or 8
        constexpr int kRPI = 64 / kPPR;             // rows per wave instruction: 4 or 8
        constexpr int kIPF = 16 / kRPI;             // instructions per 16-row fragment
        constexpr int kFragBytes = 16 * kRowS;
        constexpr int kEpiWave = G::kEpi / NWAVES / 16 * 16;
        // staging regions per wave (one fragment each).  (Batches of three fragments were 1 % slower than
        // one at a time, batches of nine through the whole 160 KiB 3 % slower.)
        constexpr int kRegions = (kEpiWave / kFragBytes) < 1 ? 1 : ((kEpiWave / kFragBytes) > 4 ? 4 : (kEpiWave / kFragBytes));
        static_assert(kFragBytes <= kEpiWave, "epilogue staging does not fit");
        unsigned char* ebuf = smem + wave * kEpiWave;
        const size_t rowBytes = (size_t)A.cout * ES;
        const size_t sliceOff = (size_t)waveGroup * kRowB;
        // byte offset, inside the row slice, of this lane's k-th 16-byte piece
        auto pieceOff = [&](int k) -> int {
            // kF16m6: pieces 0,1 = f16 hi of this lane's 16 channels; 2,3 = the 32-byte e2m3 block this lane
            // encodes (even lane group: the chunk's hi block at +64, odd: its lo block at +96) -- the kF16x3 map
            if constexpr (kSplit || kM6) return (g >> 1) * 128 + (g & 1) * 32 + (k & 1) * 16 + (k >> 1) * 64;
            else if constexpr (kM8) // pieces 0,1: f16 hi; 2: e4m3(hi); 3: e4m3(lo * 2^12) of this lane's 16 channels
                return (g >> 1) * 128 + (k < 2 ? (g & 1) * 32 + k * 16 : 64 + (k - 2) * 32 + (g & 1) * 16);
            else return g * (kRowB / 4) + k * 16;
        };
        // kF16m8, last trunk layer: the output is written in the kF16x3 layout (hi, f16 lo)
        const bool outX3 = kM8 && A.outF16x3;
        auto pieceOffX3 = [&](int k) -> int { return (g >> 1) * 128 + (g & 1) * 32 + (k & 1) * 16 + (k >> 1) * 64; };
        const int lrow = lane / kPPR;   // lane-linear view: row within an instruction
        const int lpc = lane % kPPR;    //                   piece within the row
        // global addresses = wave-uniform row base (scalar arithmetic) + one per-lane 32-bit offset
        const unsigned laneOff = (unsigned)lrow * (unsigned)rowBytes + (unsigned)lpc * 16u;
        const size_t tileOff = (row0 + (kPerm ? (size_t)0 : (size_t)fBaseE * 16)) * rowBytes + sliceOff; // (edge-packed: the table holds offsets inside the tile)
        const unsigned char* resBase = A.res + tileOff;
        unsigned char* yBase = A.y + tileOff;
        [[maybe_unused]] __amdgpu_buffer_rsrc_t resRs, yRs;
        if constexpr (COOP) {
            resRs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(resBase), 0, 0x7fffffff, 0x00027000);
            yRs = __builtin_amdgcn_make_buffer_rsrc(yBase, 0, 0x7fffffff, 0x00027000);
        }
        auto resLoad = [&](size_t off_) -> u32x4 {
            if constexpr (COOP) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(resRs, (int)off_, 0, 16));
            else return *reinterpret_cast<const u32x4*>(resBase + off_);
        };
        auto outStore = [&](size_t off_, const u32x4& v_) {
            if constexpr (COOP) __builtin_amdgcn_raw_buffer_store_b128(v_, yRs, (int)off_, 0, kCoopSameXcd ? 0 : 16);
            else *reinterpret_cast<u32x4*>(yBase + off_) = v_;
        };
        // edge-packed rows: per-lane offsets of the four rows (it = 0..3) this lane moves for fragment f
        [[maybe_unused]] const unsigned permLane = (unsigned)lpc * 16u;
        const int fBasePerm = (KS > 1) ? (wave % KS) * kMFe : PART * kMFe; // this wave's first fragment in the workgroup's row table
        auto rowOffs = [&](int f) -> u32x4 {
            if constexpr (kPerm) return *reinterpret_cast<const u32x4*>(smem + G::kRowTabOff + ((fBasePerm + f) * 4 + lrow) * 16);
            else return u32x4{0u, 0u, 0u, 0u};
        };
        // (ok, byte offset from resBase / yBase) of row it*kRPI + lrow of fragment f
        auto rowOk = [&](int f, int it, const u32x4& ro) -> bool {
            if constexpr (kPerm) return (SIZE == 2 && KS == 1 && PART * kMFe + f < EdgeRows::kMF - 1) || ro[it] != ~0u; // (two boards: only the last fragment, and what lies past the tile, has padding rows)
            else return (fBaseE + f) * 16 + it * kRPI + lrow < rowLimit;
        };
        auto rowByte = [&](int f, int it, const u32x4& ro) -> size_t {
            if constexpr (kPerm) return (size_t)(ro[it] + permLane);
            else return (size_t)(f * 16 + it * kRPI) * rowBytes + laneOff;
        };

        // Most of this wave's residual slice is requested up front (the main loop's operand registers
        // are dead), so the fragment pipeline below does not wait a global round trip per fragment.
        // (only the first kPreFrags fragments: with all eleven the epilogue spills; kernels capped at
        // 256 registers for two waves per SIMD request every fragment one pipeline stage ahead instead)
        constexpr int kPreFrags = (RES != 1 || minWavesPerSimd<MODE, SIZE, NWAVES, NFRAG, PREC>() > 1)
                                      ? 0 : (kMFe * kIPF <= 32 ? kMFe : 32 / kIPF);
        u32x4 rpre[kPreFrags ? kPreFrags : 1][kIPF];
        if constexpr (kPreFrags > 0) {
#pragma unroll
            for (int f = 0; f < kPreFrags; ++f) {
                const u32x4 ro = rowOffs(f);
#pragma unroll
                for (int it = 0; it < kIPF; ++it) {
                    rpre[f][it] = u32x4{0u, 0u, 0u, 0u};
                    if (__builtin_expect(rowOk(f, it, ro), 1))
                        rpre[f][it] = resLoad(rowByte(f, it, ro));
                }
            }
        }

        // Three stages per fragment, software-pipelined over the fragments so that no stage waits for
        // the LDS round trip of the one before it:
        //   R(f): residual pieces -> LDS image (lane-linear rows) -> this lane's MFMA-layout pieces
        //   X(f): bias, residual, ReLU, conversions; result pieces -> LDS image
        //   S(f): LDS image -> global, lane-linear rows
        // Iteration i issues S's reads for fragment i-1, R(i+1), X(i), then S's stores for i-1.  LDS
        // executes a wave's operations in order, so two staging regions suffice: the reads of
        // fragment i-1 are issued before R(i+1) overwrites their region.
        static_assert(kRegions >= 2, "the pipelined epilogue needs two staging regions per wave");
        // Where row `row_` of fragment f, 16-byte piece `piece_` of this wave's row slice, is staged.  KEEP: in the next
        // layer's image -- chunk buffer 2 * waveGroup + piece / 8, plane piece % 8, the row's entry (every fragment has
        // its own rows: no region is reused); a row past the board goes to the lane's trash slot (it is never stored,
        // and must not touch the halo row its entry formula would hit).
        [[maybe_unused]] int entLi[KEEP ? kMFe : 1], entLL[KEEP ? kMFe : 1][kIPF];
        if constexpr (KEEP) {
            static_assert(kRowB == 256 && kIPF == 4, "image staging: 256-byte row slices");
            const int slice_ = waveGroup * 2 * G::kBuf;
            // (entryOfRow for m < 128: 24 + (y + 1) * 10 + x = 34 + m + m / 9; -1: a padding row)
            auto entryByte = [&](int m_) { return m_ < G::kRows ? slice_ + (34 + m_ + ((m_ * 57) >> 9)) * 16 : -1; };
            // board row of row r_ of the workgroup's fragment fg_ (edge-packed rows: by the table's formula; a row or a
            // fragment the tile does not have: past the board)
            auto rowOfFrag = [&](int fg_, int r_) -> int {
                if constexpr (kPerm) {
                    int b_ = 0, y_ = 0, x_ = 0;
                    const bool ok_ = EdgeRows::square(fg_ < EdgeRows::kMF ? fg_ : 0, r_, b_, y_, x_) && fg_ < EdgeRows::kMF;
                    return ok_ ? y_ * 9 + x_ : G::kRows;
                } else {
                    return fg_ * 16 + r_;
                }
            };
#pragma unroll
            for (int f = 0; f < kMFe; ++f) {
                entLi[f] = entryByte(rowOfFrag(fBaseE + f, li));
#pragma unroll
                for (int it = 0; it < kIPF; ++it) entLL[f][it] = entryByte(rowOfFrag(fBaseE + f, it * kRPI + lrow));
            }
        }
        auto stageAt = [&](int f, int rowIsLi, int it, int piece_) -> unsigned char* { // row = li (MFMA layout) or it * kRPI + lrow
            if constexpr (KEEP) {
                const int e_ = rowIsLi ? entLi[f] : entLL[f][it];
                // (a padding row: every piece to the lane's own trash slot)
                return smem + (e_ < 0 ? G::kLds + lane * 16 : e_ + (piece_ >> 3) * G::kBuf + (piece_ & 7) * G::kPlaneStride);
            } else {
                return ebuf + (f % kRegions) * kFragBytes + (rowIsLi ? li : it * kRPI + lrow) * kRowS + piece_ * 16;
            }
        };
        u32x4 rpp[2][kNP];
        u32x4 tt[kIPF];
#pragma unroll
        for (int i = -1; i <= kMFe; ++i) {
            // (edge-packed rows: the row offsets of both stages are read from the table up front, so that
            // their LDS round trip overlaps the stages' own instead of stalling each in turn)
            [[maybe_unused]] u32x4 roS = u32x4{0u, 0u, 0u, 0u}, roR = u32x4{0u, 0u, 0u, 0u};
            if (i >= 1) { // ---- S(i-1), reads
                const int f = i - 1;
#pragma unroll
                for (int it = 0; it < kIPF; ++it)
                    tt[it] = *reinterpret_cast<const u32x4*>(stageAt(f, 0, it, lpc));
                roS = rowOffs(f);
            }
            if (hasRes && i + 1 < kMFe) { // ---- R(i+1)
                const int f = i + 1;
                if (f >= kPreFrags) roR = rowOffs(f);
                const u32x4 ro = roR;
#pragma unroll
                for (int it = 0; it < kIPF; ++it) {
                    u32x4 t = u32x4{0u, 0u, 0u, 0u};
                    if (f < kPreFrags) {
                        t = rpre[f < kPreFrags ? f : 0][it];
                    } else {
                        if (__builtin_expect(rowOk(f, it, ro), 1))
                            t = resLoad(rowByte(f, it, ro));
                    }
                    *reinterpret_cast<u32x4*>(stageAt(f, 0, it, lpc)) = t;
                }
#pragma unroll
                for (int k = 0; k < kNP; ++k) {
                    // (kF16m6: both lanes of a chunk read the chunk's whole lo block, pieces 2 and 3)
                    const int po = (kM6 && k >= 2) ? (g >> 1) * 128 + 96 + (k - 2) * 16 : pieceOff(k);
                    rpp[f & 1][k] = *reinterpret_cast<const u32x4*>(stageAt(f, 1, 0, po / 16));
                }
            }
            if (i >= 0 && i < kMFe) { // ---- X(i)
                const int f = i;
                float v[NFRAG * 4];
#pragma unroll
                for (int j = 0; j < NFRAG; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j * 4 + r] = fmaf(acc[f][j][r], A.accScale, bv[j * 4 + r]);
                [[maybe_unused]] unsigned char* lrowp = ebuf + (f % kRegions) * kFragBytes + li * kRowS;
                if (hasRes) {
                    const u32x4* rp = rpp[f & 1];
                    if constexpr (PREC == kFp32) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[k * 4 + i] += __uint_as_float(rp[k][i]);
                    } else if constexpr (kSplit) {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                v[k * 8 + 2 * i] += f16BitsToF32((uint16_t)(rp[k][i] & 0xffffu)) +
                                                    f16BitsToF32((uint16_t)(rp[2 + k][i] & 0xffffu));
                                v[k * 8 + 2 * i + 1] += f16BitsToF32((uint16_t)(rp[k][i] >> 16)) +
                                                        f16BitsToF32((uint16_t)(rp[2 + k][i] >> 16));
                            }
                    } else if constexpr (kM6) {
                        // lo block -> f16 (v_cvt_scalef32_pk32_f16_fp6 multiplies by the block scale).  The block's codes
                        // 0..15 (three dwords) are the even lane group's channels, 16..31 (the other three) the odd one's:
                        // a lane puts ITS three dwords first and takes the first sixteen values -- three selects on the
                        // source instead of eight on the decoded halves
                        typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
                        typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
                        const bool odd = (g & 1) != 0;
                        const uint32_t c0 = odd ? rp[2].w : rp[2].x, c1 = odd ? rp[3].x : rp[2].y, c2 = odd ? rp[3].y : rp[2].z;
                        const u32x6 blk = {c0, c1, c2, c0, c1, c2};
                        const float sc = __uint_as_float((rp[3].z & 0xffu) << 23);
                        u32x16 lh;
                        asm("v_cvt_scalef32_pk32_f16_fp6 %0, %1, %2" : "=&v"(lh) : "v"(blk), "v"(sc));
                        const uint32_t le[8] = {lh.s0, lh.s1, lh.s2, lh.s3, lh.s4, lh.s5, lh.s6, lh.s7};
                        const uint32_t hh[8] = {rp[0].x, rp[0].y, rp[0].z, rp[0].w, rp[1].x, rp[1].y, rp[1].z, rp[1].w};
                        // hi + lo in f32 as ONE v_fma_mix_f32 per value (hi * 1.0 + lo, both f16 halves converted on the
                        // fly: exact), added to the accumulator pair by one packed add
                        typedef float f32x2r __attribute__((ext_vector_type(2)));
                        float one = 1.0f;
                        asm("" : "+v"(one));
#pragma unroll
                        for (int d = 0; d < 8; ++d) { // dword d of my half = channels 2d, 2d+1
                            f32x2r rs;
                            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(rs[0]) : "v"(hh[d]), "v"(one), "v"(le[d]));
                            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(rs[1]) : "v"(hh[d]), "v"(one), "v"(le[d]));
                            const f32x2r sum = f32x2r{v[2 * d], v[2 * d + 1]} + rs;
                            v[2 * d] = sum[0];
                            v[2 * d + 1] = sum[1];
                        }
                    } else if constexpr (kM8) {
                        constexpr float kLoInv = 1.0f / (float)(1 << kM8LoShift);
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int d = 0; d < 4; ++d) { // dword d of the lo piece = channels 4d .. 4d+3
                            const f32x2 l01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)rp[3][d], false);
                            const f32x2 l23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)rp[3][d], true);
                            const uint32_t h01 = rp[d >> 1][(d & 1) * 2], h23 = rp[d >> 1][(d & 1) * 2 + 1];
                            v[4 * d + 0] += f16BitsToF32((uint16_t)(h01 & 0xffffu)) + l01[0] * kLoInv;
                            v[4 * d + 1] += f16BitsToF32((uint16_t)(h01 >> 16)) + l01[1] * kLoInv;
                            v[4 * d + 2] += f16BitsToF32((uint16_t)(h23 & 0xffffu)) + l23[0] * kLoInv;
                            v[4 * d + 3] += f16BitsToF32((uint16_t)(h23 >> 16)) + l23[1] * kLoInv;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                v[k * 8 + 2 * i] += unpackLo<PREC>(rp[k][i]);
                                v[k * 8 + 2 * i + 1] += unpackHi<PREC>(rp[k][i]);
                            }
                    }
                }
                // (the split formats fold the ReLU into the range clamp of their encoders below: one v_med3 per value)
                if (A.relu && !(kM8 || kSplit)) {
#pragma unroll
                    for (int i = 0; i < NFRAG * 4; ++i) v[i] = reluF(v[i]);
                }
                u32x4 op[kNP];
                [[maybe_unused]] const float floorV = A.relu ? 0.f : -65000.f;
                if constexpr (PREC == kFp32) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) op[k][i] = __float_as_uint(v[k * 4 + i]);
                } else if constexpr (kSplit) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            uint32_t h2, l2;
                            splitF16Pair(v[k * 8 + 2 * i], v[k * 8 + 2 * i + 1], floorV, h2, l2);
                            op[k][i] = h2;
                            op[2 + k][i] = l2;
                        }
                } else if constexpr (kM8) {
                    if (__builtin_expect(outX3, false)) {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                            uint32_t h2, l2;
                            splitF16Pair(v[k * 8 + 2 * i], v[k * 8 + 2 * i + 1], floorV, h2, l2);
                            op[k][i] = h2;
                            op[2 + k][i] = l2;
                        }
                    } else if constexpr (kM6) {
                        // f16 hi pairs and f16 copies of lo.  The lane groups (g, g^1) of a chunk trade what the
                        // other needs with v_permlane16_swap (row r of one register <-> row r+1 of the other:
                        // even rows end up with (own A, neighbour's A), odd rows with (neighbour's B, own B);
                        // profiles/r02/a_permlane_probe.txt): with A = hi, B = lo, even lanes hold the chunk's
                        // 32 hi values and odd lanes its 32 lo values, both in channel order, and the same
                        // swap on the two maxima leaves each lane the maximum of ITS block.  E8M0 exponent =
                        // exponent of the maximum minus 2 (OCP MX rule); one packed conversion per lane.
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
                        typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
                        // (scalars in plain arrays, vectors built by initialiser lists: element-wise writes to a
                        // 16-wide ext_vector in this loop compiled to compare/select chains, 2400 extra instructions)
                        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                        uint32_t hpv[8], sa[8], sb[8];
                        // running maxima of |x| and |x - hi| in f32, one v_max3_f32 with |.| modifiers per pair: rounding to
                        // f16 is monotonic, so the f16 of the maximum IS the maximum of the f16 values the block stores
                        // (masking the packed f16 pairs and taking packed integer maxima was four instructions per pair)
                        float mh = 0.f, ml = 0.f;
                        float mone = -1.0f;
                        asm("" : "+v"(mone));
                        (void)sizeof(u16x2);
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            const f32x2 x = {__builtin_amdgcn_fmed3f(v[2 * d], floorV, 65000.f),
                                             __builtin_amdgcn_fmed3f(v[2 * d + 1], floorV, 65000.f)};
                            const f16x2 h = __builtin_convertvector(x, f16x2);
                            const uint32_t hp = __builtin_bit_cast(uint32_t, h);
                            // lo = x - hi as ONE v_fma_mix_f32 per value: the f16 half of `hp` is converted on the fly (written
                            // as (float)h[i] the compiler emitted a v_cvt_f32_f16 per value in front of a plain fma)
                            f32x2 lx;
                            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lx[0]) : "v"(hp), "v"(mone), "v"(x[0]));
                            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lx[1]) : "v"(hp), "v"(mone), "v"(x[1]));
                            const uint32_t lp = __builtin_bit_cast(uint32_t, __builtin_convertvector(lx, f16x2));
                            hpv[d] = hp;
                            mh = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x[0]), __builtin_fabsf(x[1])), mh);
                            ml = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(lx[0]), __builtin_fabsf(lx[1])), ml);
                            const u32x2v sw = __builtin_amdgcn_permlane16_swap(hp, lp, false, false);
                            sa[d] = sw.x;
                            sb[d] = sw.y;
                        }
                        // f16 bit pattern -> E8M0: f16 exponent field e5 (bias 15) is the float exponent e5 + 112
                        const uint32_t mbits = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{mh, ml}, f16x2));
                        const uint32_t mhb = mbits & 0xffffu, mlb = mbits >> 16;
                        op[0] = u32x4{hpv[0], hpv[1], hpv[2], hpv[3]};
                        op[1] = u32x4{hpv[4], hpv[5], hpv[6], hpv[7]};
                        const u32x16 src = {sa[0], sa[1], sa[2], sa[3], sa[4], sa[5], sa[6], sa[7],
                                            sb[0], sb[1], sb[2], sb[3], sb[4], sb[5], sb[6], sb[7]};
                        const u32x2v mm = __builtin_amdgcn_permlane16_swap(mhb, mlb, false, false);
                        const uint32_t e5 = (mm.x > mm.y ? mm.x : mm.y) >> 10; // biased f16 exponent of my block's maximum
                        const uint32_t e8 = e5 + 110u;                           // (e5 + 112) - 2; e5 = 0 (zero / subnormal block): 2^-17
                        u32x6 blk;
                        const float sc = __uint_as_float(e8 << 23);
                        asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(blk) : "v"(src), "v"(sc));
                        op[2] = u32x4{blk.s0, blk.s1, blk.s2, blk.s3};
                        op[3] = u32x4{blk.s4, blk.s5, e8, e8}; // (the exponent twice: the consumer's split read, kSplitRead)
                    } else {
                        // two values per instruction where the ISA has a packed form (v_cvt_pk_f16_f32,
                        // v_pk_add_f32, v_pk_mul_f32); v_med3 needs no NaN-quieting of its inputs
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            f32x2 hf[2], lf[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int c = 4 * d + 2 * q;
                                const f32x2 x = {__builtin_amdgcn_fmed3f(v[c], floorV, 65000.f),
                                                 __builtin_amdgcn_fmed3f(v[c + 1], floorV, 65000.f)};
                                const f16x2 h = __builtin_convertvector(x, f16x2);
                                op[c >> 3][(c & 7) >> 1] = __builtin_bit_cast(uint32_t, h);
                                const f32x2 hx = __builtin_convertvector(h, f32x2);
                                const f32x2 lx = (x - hx) * (float)(1 << kM8LoShift);
                                hf[q] = f32x2{__builtin_amdgcn_fmed3f(hx[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(hx[1], -448.f, 448.f)};
                                lf[q] = f32x2{__builtin_amdgcn_fmed3f(lx[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(lx[1], -448.f, 448.f)};
                            }
                            int h8 = __builtin_amdgcn_cvt_pk_fp8_f32(hf[0][0], hf[0][1], 0, false);
                            h8 = __builtin_amdgcn_cvt_pk_fp8_f32(hf[1][0], hf[1][1], h8, true);
                            int l8 = __builtin_amdgcn_cvt_pk_fp8_f32(lf[0][0], lf[0][1], 0, false);
                            l8 = __builtin_amdgcn_cvt_pk_fp8_f32(lf[1][0], lf[1][1], l8, true);
                            op[2][d] = (uint32_t)h8;
                            op[3][d] = (uint32_t)l8;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) op[k][i] = packPair<PREC>(v[k * 8 + 2 * i], v[k * 8 + 2 * i + 1]);
                }
#pragma unroll
                for (int k = 0; k < kNP; ++k) {
                    if constexpr (KEEP) *reinterpret_cast<u32x4*>(stageAt(f, 1, 0, (__builtin_expect(outX3, false) ? pieceOffX3(k) : pieceOff(k)) / 16)) = op[k];
                    else *reinterpret_cast<u32x4*>(lrowp + (__builtin_expect(outX3, false) ? pieceOffX3(k) : pieceOff(k))) = op[k];
                }
            }
            if (i >= 1) { // ---- S(i-1), stores: kRPI rows x kRowB contiguous bytes per instruction
                const int f = i - 1;
                const u32x4 ro = roS;
#pragma unroll
                for (int it = 0; it < kIPF; ++it) {
#ifdef NSG_EXP_RUNTIME // bit 0 = the staged rows are kept alive, not stored
                    asm volatile("" ::"v"(tt[it]));
                    if (__builtin_expect(rowOk(f, it, ro) && !(A.exp & 1), 1))
#else
                    if (__builtin_expect(rowOk(f, it, ro), 1))
#endif
                        outStore(rowByte(f, it, ro), tt[it]);
                }
            }
        }
#ifdef NSG_DIAG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // stamp 3 = own stores issued and acknowledged
#endif
        NSG_STAMP(3);
        return;
    }

#pragma unroll
    for (int f = 0; f < G::kMF; ++f) {
        const int m = f * 16 + li;
        const size_t grow = row0 + m;
        const bool rowOk = G::kBoards ? (m < G::kRows) : (grow <= lastRow);
        if (!rowOk) continue;
        float v[NFRAG * 4];
#pragma unroll
        for (int j = 0; j < NFRAG; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = fmaf(acc[f][j][r], A.accScale, bv[j * 4 + r]);

        if constexpr (MODE == kHeads) {
            const int b = (int)(grow / 81);
            const int sq = (int)(grow - (size_t)b * 81);
            const int vc = A.valueChannels;
#pragma unroll
            for (int i = 0; i < NFRAG * 4; ++i) {
                const int n = cbase + i;
                if (n < vc) {
                    const float r = reluF(v[i]);
                    const size_t e = (size_t)b * A.vfeatStride + (size_t)sq * vc + n;
                    if constexpr (kSplit) {
                        // dense-layer input row: K index kk = sq*vc + n, same chunked hi/lo layout
                        const int kk = sq * vc + n;
                        uint16_t hb, lb;
                        splitF16(r, hb, lb);
                        unsigned char* row = A.vfeat + (size_t)b * A.vfeatStride * 4 + (size_t)(kk >> 5) * 128 + (kk & 31) * 2;
                        *reinterpret_cast<uint16_t*>(row) = hb;
                        *reinterpret_cast<uint16_t*>(row + 64) = lb;
                    } else if constexpr (PREC == kFp32) {
                        reinterpret_cast<float*>(A.vfeat)[e] = r;
                    } else {
                        reinterpret_cast<uint16_t*>(A.vfeat)[e] = toBits16<PREC>(r);
                    }
                } else if (n < vc + 27) {
                    A.policy[(size_t)b * 2187 + (size_t)(n - vc) * 81 + sq] = v[i];
                }
            }
        } else { // kDense: raw f32 partial sums of this K split (bias + ReLU belong to the consumer)
            float* out = reinterpret_cast<float*>(A.y) + (size_t)blockIdx.z * A.partStride + grow * (size_t)A.cout + cbase;
#pragma unroll
            for (int j = 0; j < NFRAG; ++j)
                *reinterpret_cast<f32x4*>(out + j * 4) =
                    f32x4{acc[f][j][0] * A.accScale, acc[f][j][1] * A.accScale, acc[f][j][2] * A.accScale, acc[f][j][3] * A.accScale};
        }
    }
}

template <int PREC, int MODE, int SIZE, int NFRAG, int NWAVES, bool HAS_RES, int MS = 1, int KS = 1, int SS = 1>
__global__ __launch_bounds__(NWAVES * 64, (minWavesPerSimd<MODE, SIZE, NWAVES, NFRAG, PREC>())) void tileKernel(
    // The fourteen dwords every conv prologue needs come first as plain arguments: the build
    // preloads them into SGPRs at wave launch (-amdgpu-kernarg-preload-count, a by-value struct
    // is not eligible), so the tile requests do not wait for a cold scalar load of the argument
    // block first.  The rest travels as a struct and is read where it is used.
    const unsigned char* x, const u32x4* w, const unsigned char* res, unsigned char* y, const float* bias,
    int kdim, int cout, int flags /* bit 0: relu, bit 1: outF16x3 */, float accScale, const ArgsTail T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Args A{x, w, bias, res, y, T.policy, T.vfeat, kdim, cout, T.totalRows, flags & 1, T.valueChannels,
                 T.vfeatStride, accScale, (flags >> 1) & 1, T.kSplits, T.partStride, T.stamps, flags >> 2};
    if constexpr (SS == 1) {
        tileBody<PREC, MODE, SIZE, NFRAG, NWAVES, HAS_RES ? 1 : 0, MS, KS>(A, smem, true);
    } else {
        // slab split: the SS waves of a channel group run different static instruction streams (their own
        // slabs of every chunk pair); every path meets the same workgroup barriers in the same order
        const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) % SS;
        if (part == 0) tileBody<PREC, MODE, SIZE, NFRAG, NWAVES, HAS_RES ? 1 : 0, MS, KS, SS, 0>(A, smem, true);
        else if (part == 1) tileBody<PREC, MODE, SIZE, NFRAG, NWAVES, HAS_RES ? 1 : 0, MS, KS, SS, 1>(A, smem, true);
        else if constexpr (SS > 2) {
            if (part == 2) tileBody<PREC, MODE, SIZE, NFRAG, NWAVES, HAS_RES ? 1 : 0, MS, KS, SS, 2>(A, smem, true);
            else tileBody<PREC, MODE, SIZE, NFRAG, NWAVES, HAS_RES ? 1 : 0, MS, KS, SS, 3>(A, smem, true);
        }
    }
}

// Persistent trunk: one launch runs every 3x3 layer (stem + 2 per residual block)
// for this workgroup's boards.  A workgroup owns whole boards, so layer l+1 needs
// only what this same workgroup wrote in layer l: there is NO inter-workgroup
// dependency, hence no grid barrier, no residency requirement and no launch
// boundary between layers.  Workgroups drift apart freely instead of being
// re-synchronised 2N+1 times per forward.
template <int PREC, int SIZE, int NFRAG, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, (minWavesPerSimd<kConv, SIZE, NWAVES, NFRAG, PREC>())) void trunkKernel(
    const Args* __restrict__ layers, int nLayers, int skewTicks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (skewTicks > 0) {
        // deliberate phase shift between workgroups (units of the 100 MHz constant clock): their HBM
        // bursts (tile loads, epilogue) then fall into each other's matrix phases
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)skewTicks * ((blockIdx.x >> 3) & 7);
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    for (int l = 0; l < nLayers; ++l) {
        const Args A = layers[l];
        // (two instantiations rather than a run-time residual flag: the merged body spills)
        if (A.res) tileBody<PREC, kConv, SIZE, NFRAG, NWAVES, 1>(A, smem, true); // the epilogue staging reuses the halo entries: re-zero
        else tileBody<PREC, kConv, SIZE, NFRAG, NWAVES, 0>(A, smem, true);
        // this layer's stores (all waves) land before any wave stages them as the
        // next layer's input: vmcnt(0) + workgroup barrier (same CU, same L1/L2 path)
        __syncthreads();
    }
}

// Cooperative trunk: every 3x3 layer of a batch whose boards are shared by SEVERAL workgroups (the K-split plans of the
// mid batches: gridDim.y workgroups per board, each a 64- or 128-channel slice) in ONE launch.  A per-layer kernel ends
// when its slowest workgroup does, 41 times per forward; here a workgroup waits only for the other workgroups OF ITS
// BOARD: member m of board b publishes flag[b][m] = l + 1 behind its layer-l stores and polls its neighbours' flags
// before layer l + 1 (flag values count on from `flagBase`, which the host advances by more than a launch's layers from
// launch to launch: what an earlier launch left in the array -- under any member count -- is below every value this one
// waits for, and the array needs no clearing between launches).  The hand-off is MI355X_MICROARCH's measured form: every store of the payload sc1 (write-through),
// every storing wave waits for its stores (vmcnt(0)), a workgroup barrier, ONE lane stores the flag (sc1); the consumer
// polls with sc1 loads, a workgroup barrier, then reads the payload with sc1 loads only (tileBody, COOP).  The flag
// of layer l also orders the write-after-read hazards of the three rotating activation buffers: a member can reach the
// epilogue of layer l + 2 (which overwrites what layer l + 1 read) only behind every neighbour's flag l + 2.
// All gridDim.x * gridDim.y workgroups must be resident (the host checks the grid against the CU count and keeps one
// such launch per device); every spin is bounded (~1 s): a member that gives up raises the host-mapped `status` and
// the launch unwinds; the host re-runs the batch on the per-layer kernels (nsg_capi.hip, teamRecover).
//
// Same-XCD hand-off (kCoopSameXcd, the default).  Write-through stores drop their lines from the XCD's L2, and a reader
// -- same XCD or not -- then fetches them from the memory side (MI355X_MICROARCH, handoff-payload: 66-73 GB/s per
// workgroup against 104-122 for lines kept in L2).  Workgroups are dealt round-robin over the eight XCDs in launch
// order, so with gridDim.x a multiple of eight (the host pads it; workgroups past the last board leave at once) every
// member of a board -- same blockIdx.x -- lands on ONE XCD, whose L2 all its CUs share: a plain store is in that L2
// when its vmcnt has counted down (the L1 is write-through), and an sc1 load (L1 bypassed, L2 served) by any CU of the
// XCD reads it there.  Placement is the hardware's habit, not a guarantee: every member publishes its XCC_ID in the top
// byte of its flag and every poll checks it against the poller's own; a mismatch raises `status` = 2 like a timed-out
// spin and the host re-runs the batch on the per-layer kernels and stops using this launch (nsg_capi.hip, teamRecover).
template <int PREC, int NFRAG, int NWAVES, int MS, int KS>
__global__ __launch_bounds__(NWAVES * 64, 1) void coopTrunkKernel(const Args* __restrict__ layers, int nLayers, int boards,
                                                                  unsigned* flags, unsigned flagBase, int* status, int faultBoard) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Geom<kConv, 1, NWAVES, 8>;
    if ((int)blockIdx.x >= boards) return; // (gridDim.x is padded to a multiple of eight)
    // a board's members: gridDim.y channel groups x gridDim.z row groups (K split AND row split: tileBody, kRowWG)
    const int members = (int)(gridDim.y * gridDim.z), me = (int)(blockIdx.z * gridDim.y + blockIdx.y);
    if ((int)blockIdx.x == faultBoard && me == 1) return; // (test hook: this member never publishes)
    unsigned xcc = 0;
    if constexpr (kCoopSameXcd) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        if (faultBoard == -2 && me == 1) xcc ^= 1u; // (test hook: a member that claims another XCD)
    }
    int* gaveUp = reinterpret_cast<int*>(smem + G::kLdsAlloc); // one int behind everything tileBody uses
    unsigned* mine = flags + (size_t)blockIdx.x * members;
    if (threadIdx.x == 0) *gaveUp = 0;
    for (int l = 0; l < nLayers; ++l) {
        // (the layer's arguments are requested BEFORE the poll: the scalar loads -- a cold line of the list every layer --
        // return while the flags are waited for, instead of in front of the first tile request)
        const Args A = layers[l];
        asm volatile("" ::"s"(A.x), "s"(A.w), "s"(A.res), "s"(A.y), "s"(A.bias), "s"(A.kdim), "s"(A.cout), "s"(A.relu), "s"(A.accScale),
                     "s"(A.outF16x3));
        if (l > 0) {
            if ((int)threadIdx.x < members && (int)threadIdx.x != me) { // one polling lane per neighbour (wave 0)
                int spins = 0;
                unsigned seen;
                while (((seen = __hip_atomic_load(mine + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffffu) < flagBase + (unsigned)l) {
                    if ((++spins & 255) == 0 &&
                        (spins > (1 << 21) || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                        __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        *gaveUp = 1;
                        seen = xcc << 24;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                if (kCoopSameXcd && (seen >> 24) != xcc) { // the neighbour runs on another XCD: its plain stores are not in this L2
                    __hip_atomic_store(status, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    *gaveUp = 1;
                }
            }
            __syncthreads(); // the other waves load behind the polling wave's match
            if (*gaveUp) return;
        }
        // (one row group per board: the workgroup's own channels of the layer before are in its LDS already -- tileBody, KEEP)
        constexpr bool kKeep = kCoopKeepOwnSlice && MS == 1;
        if (A.res) tileBody<PREC, kConv, 1, NFRAG, NWAVES, 1, MS, KS, 1, 0, true, kKeep>(A, smem, true, l > 0);
        else tileBody<PREC, kConv, 1, NFRAG, NWAVES, 0, MS, KS, 1, 0, true, kKeep>(A, smem, true, l > 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave: its write-through stores have left
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(mine + me, (flagBase + (unsigned)(l + 1)) | (xcc << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int PREC, int NFRAG, int NWAVES, int MS, int KS>
hipError_t launchCoopOne(const Args* layers, int nLayers, int boards, int cout, unsigned* flags, unsigned flagBase, int* status,
                         hipStream_t stream, int faultBoard) {
    using G = Geom<kConv, 1, NWAVES, 8>;
    constexpr bool kRowWG = (MS > 1 && KS > 1);
    constexpr int kChanGroups = NWAVES / (kRowWG ? KS : MS * KS);
    const int gy = cout / (kChanGroups * NFRAG * 16);
    if (gy < 1 || gy * kChanGroups * NFRAG * 16 != cout || gy * (kRowWG ? MS : 1) > 64) return hipErrorInvalidValue;
    auto k = coopTrunkKernel<PREC, NFRAG, NWAVES, MS, KS>;
    static std::atomic<int> attrDevMask{0}; // per kernel instantiation: devices whose attribute is set
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(attrDevMask.load() & (1 << dev))) {
        const hipError_t err = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsAlloc + 16);
        if (err != hipSuccess) return err;
        attrDevMask.fetch_or(1 << dev);
    }
    // (gridDim.x padded to a multiple of eight: blockIdx.x picks the XCD, coopTrunkKernel)
    hipLaunchKernelGGL(k, dim3((boards + 7) / 8 * 8, gy, kRowWG ? MS : 1), dim3(G::kThreads), G::kLdsAlloc + 16, stream, layers, nLayers,
                       boards, flags, flagBase, status, faultBoard);
    return hipGetLastError();
}

#ifdef NSG_EXP_RUNTIME
inline int expFlags() { const char* e = getenv("NSG_EXP_FLAGS"); return e ? atoi(e) << 2 : 0; }
#else
constexpr int expFlags() { return 0; }
#endif

template <int PREC, int MODE, int SIZE, int NFRAG, int NWAVES, int MS = 1, int KS = 1, int SS = 1>
hipError_t launchOne(const Args& a, int gridX, hipStream_t stream) {
    using G = Geom<MODE, SIZE, NWAVES, (isMx(PREC) ? (KS > 1 ? 8 : 4) : 2)>;
    constexpr bool kRowWG = (MS > 1 && KS > 1); // row groups as workgroups (grid z), see tileBody
    constexpr int kChanGroups = NWAVES / (kRowWG ? KS : MS * KS * SS);
    const int gy = a.cout / (kChanGroups * NFRAG * 16);
    if (gy < 1 || gy * kChanGroups * NFRAG * 16 != a.cout) return hipErrorInvalidValue;
    if (KS > 1) { // every chunk tile resident: at most eight, and whole pairs for every K part
        const int chunks = a.kdim * 4 / 128;
        if (chunks > 8 || chunks % (2 * KS) != 0) return hipErrorInvalidValue;
    }
    hipError_t err;
    if (MODE == kConv && a.res) {
        auto k = tileKernel<PREC, MODE, SIZE, NFRAG, NWAVES, (MODE == kConv), MS, KS, SS>;
        static std::atomic<int> attrDevMask{0}; // per kernel instantiation: devices whose attribute is set
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!(attrDevMask.load() & (1 << dev))) {
            err = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsAlloc);
            if (err != hipSuccess) return err;
            attrDevMask.fetch_or(1 << dev);
        }
        hipLaunchKernelGGL(k, dim3(gridX, gy, kRowWG ? MS : 1), dim3(G::kThreads), G::kLdsAlloc, stream, a.x, a.w, a.res, a.y, a.bias,
                           a.kdim, a.cout, (a.relu ? 1 : 0) | (a.outF16x3 ? 2 : 0) | expFlags(), a.accScale, tailOf(a));
    } else {
        auto k = tileKernel<PREC, MODE, SIZE, NFRAG, NWAVES, false, MS, KS, SS>;
        static std::atomic<int> attrDevMask{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!(attrDevMask.load() & (1 << dev))) {
            err = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsAlloc);
            if (err != hipSuccess) return err;
            attrDevMask.fetch_or(1 << dev);
        }
        hipLaunchKernelGGL(k, dim3(gridX, gy, MODE == kDense ? a.kSplits : (kRowWG ? MS : 1)), dim3(G::kThreads), G::kLdsAlloc, stream,
                           a.x, a.w, a.res, a.y, a.bias, a.kdim, a.cout, (a.relu ? 1 : 0) | (a.outF16x3 ? 2 : 0) | expFlags(), a.accScale, tailOf(a));
    }
    return hipGetLastError();
}

template <int PREC, int SIZE, int NFRAG, int NWAVES>
hipError_t launchTrunkOne(const Args* layers, int nLayers, int gridX, hipStream_t stream) {
    using G = Geom<kConv, SIZE, NWAVES, (isMx(PREC) ? 4 : 2)>;
    auto k = trunkKernel<PREC, SIZE, NFRAG, NWAVES>;
    hipError_t err = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsAlloc);
    if (err != hipSuccess) return err;
    static const int skew = [] { const char* e = getenv("NSG_TRUNK_SKEW_US"); return e ? atoi(e) * 100 : 0; }();
    hipLaunchKernelGGL(k, dim3(gridX, 1), dim3(G::kThreads), G::kLdsAlloc, stream, layers, nLayers, skew);
    return hipGetLastError();
}

// Per-precision entry points (one translation unit each).
template <int PREC>
hipError_t launchTrunkPrec(const Args* layers, int nLayers, int batch, const ConvPlan& p, hipStream_t stream) {
    const int gx = (batch + p.nb - 1) / p.nb;
#define NSG_CASE(NB_, NW_) \
    if (p.nb == NB_ && p.nfrag == 4 && p.nwaves == NW_) return launchTrunkOne<PREC, NB_, 4, NW_>(layers, nLayers, gx, stream);
    NSG_CASE(2, 4) NSG_CASE(2, 3) NSG_CASE(1, 4) NSG_CASE(1, 3)
#undef NSG_CASE
    return hipErrorInvalidValue;
}
template <int PREC>
hipError_t launchConvPrec(const Args& a, int batch, const ConvPlan& p, hipStream_t stream) {
    const int gx = (batch + p.nb - 1) / p.nb;
#define NSG_CASE(NB_, NF_, NW_) \
    if (p.nb == NB_ && p.nfrag == NF_ && p.nwaves == NW_) return launchOne<PREC, kConv, NB_, NF_, NW_>(a, gx, stream);
    NSG_CASE(2, 4, 4) NSG_CASE(2, 4, 3) NSG_CASE(2, 4, 2) NSG_CASE(2, 4, 1)
    NSG_CASE(1, 4, 4) NSG_CASE(1, 4, 3) NSG_CASE(1, 4, 2) NSG_CASE(1, 4, 1)
    // small-batch tiles: fewer channels per wave, four waves share one input image
    if (p.msplit == 2 && p.nb == 1 && p.nfrag == 2 && p.nwaves == 4) return launchOne<PREC, kConv, 1, 2, 4, 2>(a, gx, stream);
    NSG_CASE(2, 2, 4) NSG_CASE(2, 1, 4) NSG_CASE(1, 2, 4) NSG_CASE(1, 1, 4)
#undef NSG_CASE
    return hipErrorInvalidValue;
}
template <int PREC>
hipError_t launchHeadsPrec(const Args& a, hipStream_t stream) {
    constexpr int kMF = 2; // 32 rows per one-wave workgroup: ~5 waves per CU hide the per-chunk latencies
    const int gx = (a.totalRows + kMF * 16 - 1) / (kMF * 16);
    return launchOne<PREC, kHeads, kMF, 4, 1>(a, gx, stream);
}
template <int PREC>
hipError_t launchDensePrec(const Args& a, hipStream_t stream) {
    constexpr int kMF = 2; // 32 rows per workgroup
    const int gx = (a.totalRows + kMF * 16 - 1) / (kMF * 16);
    return launchOne<PREC, kDense, kMF, 4, 1>(a, gx, stream);
}

hipError_t launchConvFp32(const Args& a, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchConvFp16(const Args& a, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchConvBf16(const Args& a, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchHeadsFp32(const Args& a, hipStream_t s);
hipError_t launchHeadsFp16(const Args& a, hipStream_t s);
hipError_t launchHeadsBf16(const Args& a, hipStream_t s);
hipError_t launchDenseFp32(const Args& a, hipStream_t s);
hipError_t launchDenseFp16(const Args& a, hipStream_t s);
hipError_t launchDenseBf16(const Args& a, hipStream_t s);
hipError_t launchConvF16x3(const Args& a, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchConvF16m8(const Args& a, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchTrunkF16m8(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchConvF16m6(const Args& a, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchTrunkF16m6(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchCoopTrunkF16m6(const Args* layers, int n, int batch, int cout, const ConvPlan& p, unsigned* flags, unsigned flagBase,
                                int* status, hipStream_t s, int faultBoard);
hipError_t launchTrunkFp32(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchTrunkFp16(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchTrunkBf16(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchTrunkF16x3(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s);
hipError_t launchHeadsF16x3(const Args& a, hipStream_t s);
hipError_t launchDenseF16x3(const Args& a, hipStream_t s);

} // namespace tile
} // namespace nsg

#endif
