// tile_launch.hip -- host side of the MFMA tile kernels: tile-plan choice,
// precision dispatch, weight packing, and the small value-output kernel.
#include "mfma_tile.h"

#include <algorithm>
#include <cmath>
#include <stdlib.h>
#include <string.h>

namespace nsg {

ConvTuning readConvTuning() {
    ConvTuning t;
    if (const char* e = getenv("NSG_CONV_NB")) t.nb = atoi(e);
    if (const char* e = getenv("NSG_CONV_NWAVES")) t.nwaves = atoi(e);
    if (const char* e = getenv("NSG_CONV_NFRAG")) t.nfrag = atoi(e);
    if (const char* e = getenv("NSG_CONV_MSPLIT")) t.msplit = atoi(e);
    if (const char* e = getenv("NSG_ROWSPLIT8_MAX_BATCH")) t.rowsplit8Max = atoi(e);
    if (const char* e = getenv("NSG_SPLIT_BATCH")) t.splitBatch = atoi(e);
    if (const char* e = getenv("NSG_SPLIT_BATCH_MAX")) t.splitBatchMax3 = atoi(e);
    if (const char* e = getenv("NSG_SLAB_SPLIT")) t.slabSplit = atoi(e);
    if (const char* e = getenv("NSG_KSPLIT3")) t.ksplit3 = atoi(e);
    return t;
}

ConvPlan chooseConvPlan(int batch, int cout, int computeUnits, const ConvTuning& tune) {
    ConvPlan p;
    const int groups = cout / (kNfrag * 16); // 64-channel weight groups
    // Full tiles: 4 fragments (64 channels) per wave, widest workgroup that divides
    // the channel groups (its waves share one LDS image of the input tile).
    p.nfrag = kNfrag;
    // A full tile's wave needs the whole register file of its SIMD, so a CU hosts
    // 4 waves: one 4-wave workgroup or two 2-wave workgroups fill it, a 3-wave
    // workgroup leaves a SIMD idle (40x384: 2-wave groups +17 % over 3-wave groups).
    p.nwaves = (groups % 4 == 0) ? 4 : (groups % 2 == 0) ? 2 : (groups % 3 == 0) ? 3 : 1;
    // Two boards per workgroup waste less of the last 16-row fragment (162 -> 176
    // rows vs 81 -> 96) but halve the grid: use them once the grid still covers
    // most of the chip.
    const int simds = computeUnits * 4;
    auto waves = [&](int nb, int nfrag) { return ((batch + nb - 1) / nb) * (cout / (16 * nfrag)); };
    p.nb = (waves(2, 4) * 4 >= simds * 3) ? 2 : 1;
    // Small batches: a wave's run time is set by fragments-per-wave x K, so when
    // the grid leaves SIMDs idle give each wave fewer channels (2 or 1 fragments,
    // four waves per workgroup) until the chip is covered.
    if (waves(p.nb, 4) * 4 < simds * 3) {
        for (int nf : {2, 1}) {
            for (int nb : {2, 1}) {
                if (cout % (4 * nf * 16) != 0) continue;
                if (waves(nb, nf) * 4 >= simds * 3 || (nf == 1 && nb == 1)) {
                    p.nb = nb; p.nfrag = nf; p.nwaves = 4;
                    goto chosen;
                }
            }
        }
    }
chosen:
    // A small-tile plan whose grid needs a second round of workgroups (more workgroups than CUs:
    // the CUs that get two take twice as long as the rest) loses to one full one-board tile per
    // board when those tiles fit in one round: B = 129..191 on 256 CUs (129: 2.13 -> 1.9 ms).
    if (p.nfrag < 4 && tune.nfrag == 0 && tune.nb == 0) {
        const int perBoard = cout / (16 * p.nfrag * 4); // workgroups per tile of the small plan
        const int nwg = ((batch + p.nb - 1) / p.nb) * perBoard;
        if (nwg > computeUnits && batch <= computeUnits && 2 * batch > computeUnits) {
            p.nb = 1; p.nfrag = 4;
            p.nwaves = (groups % 4 == 0) ? 4 : (groups % 2 == 0) ? 2 : (groups % 3 == 0) ? 3 : 1;
        }
    }
    const int kEnvNb = tune.nb, kEnvNwaves = tune.nwaves, kEnvNfrag = tune.fullTilesOnly ? 4 : tune.nfrag;
    if (kEnvNb == 1 || kEnvNb == 2) p.nb = kEnvNb;
    if (kEnvNfrag) {
        const int v = kEnvNfrag;
        if (v == 1 || v == 2) { p.nfrag = v; p.nwaves = 4; }
        if (v == 4) { p.nfrag = 4; p.nwaves = (groups % 4 == 0) ? 4 : (groups % 2 == 0) ? 2 : (groups % 3 == 0) ? 3 : 1; }
    }
    if (kEnvNwaves >= 1 && kEnvNwaves <= 4 && groups % kEnvNwaves == 0 && p.nfrag == kNfrag) p.nwaves = kEnvNwaves;
    // One board, one fragment per wave: every wave reads every row fragment from LDS for one MFMA
    // each (LDS-bound).  Two wave groups on half the rows each with two fragments per wave cover
    // the same 64 channels per workgroup with half the LDS reads.
    if (p.nb == 1 && p.nfrag == 1 && p.nwaves == 4 && tune.msplit != 1 && kEnvNfrag == 0) {
        p.nfrag = 2;
        p.msplit = 2;
    }
    return p;
}

hipError_t launchConv3x3(const void* x, const void* wfrag, const float* bias,
                         const void* residual, void* y, int batch, int cin,
                         int cout, int relu, float accScale, int prec,
                         const ConvPlan& plan, hipStream_t stream,
                         unsigned long long* stamps, bool outF16x3) {
    if (batch <= 0 || cin % inputChannelGranule(prec) != 0 || cout % 64 != 0)
        return hipErrorInvalidValue;
    tile::Args a{};
    a.x = (const unsigned char*)x;
    a.w = (const tile::u32x4*)wfrag;
    a.bias = bias;
    a.res = (const unsigned char*)residual;
    a.y = (unsigned char*)y;
    a.kdim = cin;
    a.cout = cout;
    a.totalRows = batch * 81;
    a.relu = relu;
    a.accScale = accScale;
    a.stamps = stamps;
    a.outF16x3 = outF16x3 ? 1 : 0;
    switch (prec) {
    case kF16m8: return tile::launchConvF16m8(a, batch, plan, stream);
    case kF16m6: return tile::launchConvF16m6(a, batch, plan, stream);
    case kFp32: return tile::launchConvFp32(a, batch, plan, stream);
    case kFp16: return tile::launchConvFp16(a, batch, plan, stream);
    case kBf16: return tile::launchConvBf16(a, batch, plan, stream);
    case kF16x3: return tile::launchConvF16x3(a, batch, plan, stream);
    }
    return hipErrorInvalidValue;
}

size_t trunkLayerBytes() { return sizeof(tile::Args); }
size_t trunkLayerStampsOffset() { return offsetof(tile::Args, stamps); }

void fillTrunkLayer(void* hostLayers, int index, const void* x, const void* wfrag,
                    const float* bias, const void* residual, void* y, int cin,
                    int cout, int relu, float accScale, bool outF16x3) {
    tile::Args a{};
    a.outF16x3 = outF16x3 ? 1 : 0;
    a.x = (const unsigned char*)x;
    a.w = (const tile::u32x4*)wfrag;
    a.bias = bias;
    a.res = (const unsigned char*)residual;
    a.y = (unsigned char*)y;
    a.kdim = cin;
    a.cout = cout;
    a.relu = relu;
    a.accScale = accScale;
    memcpy((unsigned char*)hostLayers + (size_t)index * sizeof(tile::Args), &a, sizeof(a));
}

bool canRunTrunk(int cout, const ConvPlan& plan) {
    return plan.nfrag == kNfrag && plan.msplit == 1 && plan.ksplit == 1 && plan.sslab == 1 && cout == plan.nwaves * 64 &&
           (plan.nwaves == 4 || plan.nwaves == 3);
}

hipError_t launchTrunk(const void* devLayers, int nLayers, int batch, int prec,
                       const ConvPlan& plan, hipStream_t stream) {
    if (batch <= 0 || nLayers <= 0) return hipErrorInvalidValue;
    const tile::Args* L = (const tile::Args*)devLayers;
    switch (prec) {
    case kFp32: return tile::launchTrunkFp32(L, nLayers, batch, plan, stream);
    case kFp16: return tile::launchTrunkFp16(L, nLayers, batch, plan, stream);
    case kBf16: return tile::launchTrunkBf16(L, nLayers, batch, plan, stream);
    case kF16x3: return tile::launchTrunkF16x3(L, nLayers, batch, plan, stream);
    case kF16m8: return tile::launchTrunkF16m8(L, nLayers, batch, plan, stream);
    case kF16m6: return tile::launchTrunkF16m6(L, nLayers, batch, plan, stream);
    }
    return hipErrorInvalidValue;
}

// The K-split plans of an f16m6 evaluator.  *firstSeparate: the stem has fewer chunk pairs than the plan splits K by and
// runs another kernel (its own launch, ahead of the cooperative one, which then starts at the first residual layer).
bool canRunCoopTrunk(int cout, int stemKdim, int prec, const ConvPlan& plan, bool* firstSeparate) {
    if (prec != kF16m6 || plan.nb != 1 || plan.nfrag != kNfrag || plan.sslab != 1) return false;
    const bool rows = plan.msplit == 1 || plan.msplit == 2 || plan.msplit == 3 || plan.msplit == 6;
    bool sep = false, ok = false;
    if (cout == 256 && plan.nwaves == 4 && plan.ksplit == 2 && plan.msplit == 1) { ok = true; sep = !(stemKdim % 128 == 0 && stemKdim <= 256); }
    else if (cout == 256 && plan.nwaves == 4 && plan.ksplit == 4 && rows) { ok = true; sep = !(stemKdim == 256); }
    else if (cout == 192 && plan.nwaves == 3 && plan.ksplit == 3 && rows) { ok = true; sep = !(stemKdim == 192); }
    if (firstSeparate) *firstSeparate = sep;
    return ok;
}

int coopMembers(int cout, const ConvPlan& plan) {
    const bool rowWG = plan.msplit > 1 && plan.ksplit > 1;
    const int chanGroups = plan.nwaves / (rowWG ? plan.ksplit : plan.msplit * plan.ksplit);
    return cout / (chanGroups * plan.nfrag * 16) * (rowWG ? plan.msplit : 1);
}

// Every workgroup of the cooperative launch resident at once.  Same-XCD hand-off (tile::kCoopSameXcd): blockIdx.x picks
// the XCD (workgroups are dealt round-robin, gridDim.x is padded to a multiple of eight), so the members of
// ceil(boards / 8) boards must fit ONE XCD's share of the CUs.
bool coopFits(int boards, int members, int computeUnits) {
    if (tile::kCoopSameXcd) return (long)((boards + 7) / 8) * members <= computeUnits / 8;
    return (long)boards * members <= computeUnits;
}

hipError_t launchCoopTrunk(const void* devLayers, int nLayers, int batch, int cout, int prec, const ConvPlan& plan,
                           unsigned* flags, unsigned flagBase, int* status, hipStream_t stream, int faultBoard) {
    if (batch <= 0 || nLayers <= 0 || !flags || !status || prec != kF16m6 || flagBase + (unsigned)nLayers + 1 >= (1u << 24))
        return hipErrorInvalidValue;
    return tile::launchCoopTrunkF16m6((const tile::Args*)devLayers, nLayers, batch, cout, plan, flags, flagBase, status, stream, faultBoard);
}

hipError_t launchHeads(const void* x, const void* wfrag, const float* bias,
                       float* policy, void* vfeat, int batch, int channels,
                       int coutPadded, int valueChannels, int vfeatStride,
                       float accScale, int prec, hipStream_t stream) {
    if (batch <= 0 || (channels * elemSize(prec)) % 128 != 0 || coutPadded % 64 != 0 ||
        valueChannels + 27 > coutPadded || vfeatStride < 81 * valueChannels)
        return hipErrorInvalidValue;
    tile::Args a{};
    a.x = (const unsigned char*)x;
    a.w = (const tile::u32x4*)wfrag;
    a.bias = bias;
    a.policy = policy;
    a.vfeat = (unsigned char*)vfeat;
    a.kdim = channels;
    a.cout = coutPadded;
    a.totalRows = batch * 81;
    a.valueChannels = valueChannels;
    a.vfeatStride = vfeatStride;
    a.accScale = accScale;
    switch (prec) {
    case kFp32: return tile::launchHeadsFp32(a, stream);
    case kFp16: return tile::launchHeadsFp16(a, stream);
    case kBf16: return tile::launchHeadsBf16(a, stream);
    case kF16x3: return tile::launchHeadsF16x3(a, stream);
    }
    return hipErrorInvalidValue;
}

int denseSplits(int kdim, int prec) {
    const int nkc = kdim / chunkChannels(prec);
    for (int s = 9; s > 1; --s)
        if (nkc % s == 0) return s;
    return 1;
}

hipError_t launchDense(const void* x, const void* wfrag, const float* bias,
                       float* y, int rows, int kdim, int cout, size_t partStride,
                       float accScale, int prec, hipStream_t stream) {
    if (rows <= 0 || (kdim * elemSize(prec)) % 128 != 0 || cout % 64 != 0)
        return hipErrorInvalidValue;
    tile::Args a{};
    a.x = (const unsigned char*)x;
    a.w = (const tile::u32x4*)wfrag;
    a.bias = bias;
    a.y = (unsigned char*)y;
    a.kdim = kdim;
    a.cout = cout;
    a.totalRows = rows;
    a.relu = 0;
    a.accScale = accScale;
    a.kSplits = denseSplits(kdim, prec);
    a.partStride = partStride;
    switch (prec) {
    case kFp32: return tile::launchDenseFp32(a, stream);
    case kFp16: return tile::launchDenseFp16(a, stream);
    case kBf16: return tile::launchDenseBf16(a, stream);
    case kF16x3: return tile::launchDenseF16x3(a, stream);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Value MLP layer 2: one wave per board, 64-lane shuffle reduction.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void valueOutKernel(
    const float* __restrict__ h, const float* __restrict__ b1, int nsplit, size_t partStride,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ value,
    float* __restrict__ draw, int batch, int hidden) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    float s0 = 0.f, s1 = 0.f;
    for (int j = lane; j < hidden; j += 64) {
        // layer 1: the K splits' partial sums in a fixed order, + bias, ReLU (all slices
        // requested before the first is added: one memory round trip, not nsplit)
        float part[9];
#pragma unroll
        for (int z = 0; z < 9; ++z) part[z] = z < nsplit ? h[(size_t)z * partStride + (size_t)b * hidden + j] : 0.f;
        float hv = b1[j];
#pragma unroll
        for (int z = 0; z < 9; ++z) hv += part[z];
        hv = fmaxf(hv, 0.f);
        s0 = fmaf(hv, w2[j], s0);
        s1 = fmaf(hv, w2[hidden + j], s1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
    }
    if (lane == 0) {
        const float o0 = s0 + b2[0];
        const float o1 = s1 + b2[1];
        value[b] = 0.5f * (tanhf(o0) + 1.0f);
        draw[b] = 1.0f / (1.0f + expf(-o1));
    }
}
} // namespace

hipError_t launchValueOut(const float* h, const float* b1, int nsplit, size_t partStride,
                          const float* w2, const float* b2,
                          float* value, float* draw, int batch, int hidden,
                          hipStream_t stream) {
    if (batch <= 0 || nsplit < 1 || nsplit > 9) return hipErrorInvalidValue;
    hipLaunchKernelGGL(valueOutKernel, dim3((batch + 3) / 4), dim3(256), 0, stream,
                       h, b1, nsplit, partStride, w2, b2, value, draw, batch, hidden);
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void gatherLogitsKernel(
    const float* __restrict__ policy, const uint16_t* __restrict__ idx,
    const uint32_t* __restrict__ offsets, float* __restrict__ out, int batch, int softmax) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const uint32_t lo = offsets[b], hi = offsets[b + 1];
    const float* row = policy + (size_t)b * 2187;
    if (!softmax) {
        for (uint32_t i = lo + lane; i < hi; i += 64) out[i] = row[idx[i] < 2187 ? idx[i] : 0];
        return;
    }
    // softmax(T=1) over this position's moves: max, sum of exp, normalise (f32, like the host's)
    float m = -INFINITY;
    for (uint32_t i = lo + lane; i < hi; i += 64) m = fmaxf(m, row[idx[i] < 2187 ? idx[i] : 0]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float sum = 0.f;
    for (uint32_t i = lo + lane; i < hi; i += 64) {
        const float e = expf(row[idx[i] < 2187 ? idx[i] : 0] - m);
        out[i] = e;
        sum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
    for (uint32_t i = lo + lane; i < hi; i += 64) out[i] *= inv;
}
} // namespace

hipError_t launchGatherLogits(const float* policy, const uint16_t* idx, const uint32_t* offsets,
                              float* out, int batch, int softmax, hipStream_t stream) {
    if (batch <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gatherLogitsKernel, dim3((batch + 3) / 4), dim3(256), 0, stream,
                       policy, idx, offsets, out, batch, softmax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Host-side weight packing.  Record (q, nf, lane) holds, for MFMA row
// rho = lane & 15 and lane group g = lane >> 4 of output fragment nf:
//   out channel n = (nf / 4) * 64 + (rho >> 2) * 16 + (nf % 4) * 4 + (rho & 3)
//   record      q = (kc*taps + tap)*2 + s
//   f32  : 4 values, input channel kc*32 + s*16 + 4*g + i          (i = 0..3)
//   16b  : 8 values, input channel kc*64 + s*32 + 8*g + i          (i = 0..7)
//   f16x3: 8 values, input channel kc*32 + 8*g + i; s = 0: hi, 1: lo
// Sixteen zero records are appended so the kernel's prefetch (up to 16 records ahead for
// small-batch tiles) never reads past the allocation.
// ---------------------------------------------------------------------------
static inline uint16_t hostF32ToF16(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static inline uint16_t hostF32ToBf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u); // round to nearest even
    return (uint16_t)(u >> 16);
}

size_t tileWeightRecords(int taps, int kdim, int cout, int prec) {
    const int nkc = kdim / chunkChannels(prec);
    const size_t nft = (size_t)(cout / 16);
    if (prec == kF16m6) // packed MX slabs: per tap and chunk pair nft x (1024 + 1024 + 1600) bytes (mfma_tile.h, kPackX)
        return (size_t)(nkc / 2) * taps * nft * 3648 / 16 + 16 * nft * 64;
    return ((size_t)nkc * recordsPerChunk(taps, prec) + 16) * nft * 64; // + 16 zero records: deepest prefetch
}

namespace {
// OCP e4m3 (bias 7, max 448, no infinities), round to nearest even, saturating.
uint8_t hostF32ToE4m3(float v) {
    const uint8_t sign = std::signbit(v) ? 0x80 : 0;
    float a = std::fabs(v);
    if (!(a == a)) return sign | 0x7f;
    if (a >= 448.f) return sign | 0x7e;
    if (a < 0x1p-10f) return sign; // below half the smallest subnormal (2^-9)
    int e;
    (void)std::frexp(a, &e); // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;          // a in [2^ex, 2^(ex+1))
    if (ex < -6) ex = -6;    // subnormal range: step 2^-9
    const float step = std::ldexp(1.f, ex - 3);
    float q = std::nearbyint(a / step); // ties to even (default rounding mode)
    float r = q * step;
    if (r >= 448.f) return sign | 0x7e;
    // re-derive exponent after rounding (may have carried)
    int e2;
    (void)std::frexp(r, &e2);
    int ex2 = e2 - 1;
    if (r < 0x1p-6f) { // subnormal
        return sign | (uint8_t)std::lrint(r / 0x1p-9f);
    }
    const int mant = (int)std::lrint(r / std::ldexp(1.f, ex2 - 3)) - 8;
    return sign | (uint8_t)(((ex2 + 7) << 3) | (mant & 7));
}

// e2m3 (fp6: 1 sign, 2 exponent, 3 mantissa bits, bias 1, max 7.5), round to nearest even, saturating;
// the value is already divided by its block scale.
uint8_t hostF32ToE2m3(float v) {
    const uint8_t sign = std::signbit(v) ? 0x20 : 0;
    float a = std::fabs(v);
    if (!(a == a) || a >= 7.5f) return sign | 0x1f;
    int ex = 0; // a in [2^ex, 2^(ex+1)), clamped to the format's three binades; below 1: subnormal step 1/8
    if (a >= 4.f) ex = 2; else if (a >= 2.f) ex = 1;
    const float step = std::ldexp(1.f, ex - 3);
    const float r = std::nearbyint(a / step) * step; // ties to even
    if (r >= 7.5f) return sign | 0x1f;
    if (r < 1.f) return sign | (uint8_t)std::lrint(r * 8.f);
    int e2 = r >= 4.f ? 2 : (r >= 2.f ? 1 : 0);
    const int mant = (int)std::lrint(r / std::ldexp(1.f, e2 - 3)) - 8;
    return sign | (uint8_t)(((e2 + 1) << 3) | (mant & 7));
}

// The MX operand of one lane in the kF16m6 layout: 32 values -> 24 bytes of e2m3 codes (value j at
// bits [6j, 6j+6)), then the E8M0 exponent of the block (OCP MX rule: floor(log2(max|v|)) - 2), 7 bytes pad.
void packE2m3Block(const float* v, unsigned char* out32) {
    float maxAbs = 0.f;
    for (int j = 0; j < 32; ++j) maxAbs = std::fmax(maxAbs, std::fabs(v[j]));
    int e8 = 0;
    if (maxAbs > 0.f && std::isfinite(maxAbs)) {
        int e;
        (void)std::frexp(maxAbs, &e); // maxAbs in [2^(e-1), 2^e)
        e8 = std::min(254, std::max(1, e - 1 - 2 + 127));
    }
    memset(out32, 0, 32);
    const float inv = std::ldexp(1.f, 127 - e8);
    for (int j = 0; j < 32; ++j) {
        const unsigned code = e8 ? hostF32ToE2m3(v[j] * inv) : 0u;
        const int bit = 6 * j;
        out32[bit / 8] |= (unsigned char)(code << (bit % 8));
        if (bit % 8 > 2) out32[bit / 8 + 1] |= (unsigned char)(code >> (8 - bit % 8));
    }
    out32[24] = (unsigned char)e8;
}

// kF16m8 / kF16m6 conv weights (see kernels.h): per chunk pair (A, B) and tap t the slabs A.m_t, B.m_t, X_t
// in stream order.
void packTileWeightsM8(WeightGetter get, const void* ctx, int taps, int kReal, int kdim, int cout,
                       float scale, unsigned char* out, bool fp6) {
    const int npairs = kdim / 64;
    const int nft = cout / 16;
    // per tap: [main A: nft KiB][main B: nft KiB][MX slab: kF16m8 2 nft KiB; kF16m6 nft/4 groups of 6400 bytes =
    // 4 x 1 KiB code dwords 0-3, 4 x 512 B code dwords 4-5, 256 B exponent dwords (byte j = fragment j of the group)]
    const size_t mainSet = (size_t)nft * 1024, xSet = (size_t)nft * (fp6 ? 1600 : 2048), tapStride = 2 * mainSet + xSet;
    auto chan = [](int nf, int rho) {
        return (nf / kNfrag) * kNfrag * 16 + (rho >> 2) * 4 * kNfrag + (nf % kNfrag) * 4 + (rho & 3);
    };
    auto wval = [&](int n, int k, int t) { return (k < kReal) ? get(ctx, n, k, t) * scale : 0.f; };
    for (int cp = 0; cp < npairs; ++cp) {
        for (int t = 0; t < taps; ++t) {
            unsigned char* base = out + ((size_t)cp * taps + t) * tapStride;
            for (int half = 0; half < 2; ++half) { // main slabs: w_hi of chunk A, then of chunk B, 8 f16 per lane
                const int c = 2 * cp + half;
                for (int nf = 0; nf < nft; ++nf)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rho = lane & 15, g = lane >> 4, n = chan(nf, rho);
                        unsigned char* rec = base + half * mainSet + ((size_t)nf * 64 + lane) * 16;
                        for (int i = 0; i < 8; ++i) {
                            const _Float16 h = (_Float16)wval(n, c * 32 + 8 * g + i, t);
                            memcpy(rec + i * 2, &h, 2);
                        }
                    }
            }
            unsigned char* xs = base + 2 * mainSet;
            if (fp6) { // MX slab, kF16m6: k-group g = (chunk g>>1, term g&1), one 24-byte e2m3 block + exponent per lane
                for (int nf = 0; nf < nft; ++nf)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rho = lane & 15, g = lane >> 4, n = chan(nf, rho);
                        const int c = 2 * cp + (g >> 1);
                        float v32[32];
                        for (int j = 0; j < 32; ++j) {
                            // term 0: w_lo against the x_hi block; term 1: the copy of w_hi against the x_lo block
                            const float v = wval(n, c * 32 + j, t);
                            const _Float16 h = (_Float16)v;
                            v32[j] = (g & 1) ? (float)h : v - (float)h;
                        }
                        unsigned char blk[32];
                        packE2m3Block(v32, blk);
                        unsigned char* grp = xs + (size_t)(nf / kNfrag) * 6400;
                        const int j4 = nf % kNfrag;
                        memcpy(grp + (size_t)j4 * 1024 + lane * 16, blk, 16);
                        memcpy(grp + 4096 + (size_t)j4 * 512 + lane * 8, blk + 16, 8);
                        grp[6144 + lane * 4 + j4] = blk[24];
                    }
            } else
            for (int nf = 0; nf < nft; ++nf) // MX slab: k-group g = (chunk g>>1, term g&1), 32 fp8 per lane
                for (int h16 = 0; h16 < 2; ++h16)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rho = lane & 15, g = lane >> 4, n = chan(nf, rho);
                        const int c = 2 * cp + (g >> 1);
                        unsigned char* rec = xs + (((size_t)nf * 2 + h16) * 64 + lane) * 16;
                        for (int i = 0; i < 16; ++i) {
                            const float v = wval(n, c * 32 + 16 * h16 + i, t);
                            const _Float16 h = (_Float16)v;
                            const float lo = v - (float)h;
                            rec[i] = (g & 1) ? hostF32ToE4m3(std::ldexp((float)h, kM8WHiShift))
                                             : hostF32ToE4m3(std::ldexp(lo, kM8WLoShift));
                        }
                    }
        }
    }
}
} // namespace

void packTileWeights(WeightGetter get, const void* ctx, int taps, int kReal,
                     int kdim, int cout, int prec, float scale, void* dst) {
    const int kcCh = chunkChannels(prec);
    const int nkc = kdim / kcCh;
    const int nft = cout / 16;
    const int spt = recordsPerTap(prec);
    const int per = (prec == kFp32) ? 4 : 8; // values per record
    unsigned char* out = (unsigned char*)dst;
    memset(out, 0, tileWeightRecords(taps, kdim, cout, prec) * 16);
    if (isMx(prec)) {
        packTileWeightsM8(get, ctx, taps, kReal, kdim, cout, scale, out, prec == kF16m6);
        return;
    }
    for (int c = 0; c < nkc; ++c)
        for (int t = 0; t < taps; ++t)
            for (int s = 0; s < spt; ++s) {
                const size_t q = ((size_t)c * taps + t) * spt + s;
                for (int nf = 0; nf < nft; ++nf)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rho = lane & 15, g = lane >> 4;
                        const int n = (nf / kNfrag) * kNfrag * 16 + (rho >> 2) * 4 * kNfrag +
                                      (nf % kNfrag) * 4 + (rho & 3);
                        unsigned char* rec = out + ((q * nft + nf) * 64 + lane) * 16;
                        for (int i = 0; i < per; ++i) {
                            // kF16x3: both records cover the chunk's 32 channels
                            const int k = (prec == kF16x3) ? c * kcCh + per * g + i
                                                           : c * kcCh + s * (kcCh / 2) + per * g + i;
                            const float v = (k < kReal) ? get(ctx, n, k, t) * scale : 0.f;
                            if (prec == kFp32) {
                                memcpy(rec + i * 4, &v, 4);
                            } else if (prec == kF16x3) {
                                // records: w_hi (products with x_hi and x_lo), w_lo (with x_hi)
                                const _Float16 h = (_Float16)v;
                                const _Float16 l = (_Float16)(v - (float)h);
                                const _Float16 pick = (s == 1) ? l : h;
                                memcpy(rec + i * 2, &pick, 2);
                            } else {
                                const uint16_t hb = (prec == kFp16) ? hostF32ToF16(v) : hostF32ToBf16(v);
                                memcpy(rec + i * 2, &hb, 2);
                            }
                        }
                    }
            }
}

} // namespace nsg
