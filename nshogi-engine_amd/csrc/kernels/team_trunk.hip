// team_trunk.hip -- the whole 3x3 trunk of up to eight boards in ONE persistent launch, for the smallest
// batches (the engine's leaf batches at the start of a search, batch-1 analysis).
//
// Why: at one board a 3x3 layer is 2.36 MB of weights against 15 MFLOP -- a weight-bandwidth problem.  The
// per-layer tile kernels give a board to 24 workgroups that each stream 622 KB of records through ONE L1 for 108
// MFMAs per wave, behind a launch boundary, a kernel-start latency and a tile round trip per layer: 14.5 us per
// layer.  Here a board belongs to a TEAM of 32 workgroups (one per CU; blocks b and b + 8 share an XCD under the
// observed round-robin placement, so a team is the blocks of one residue mod 8 -- speed only, nothing depends on
// it).  Member (j, h) computes 16 output channels (weight fragment j) of one half of the board's rows: its
// eight waves split K by 32-channel chunk, so a wave's share of a layer's weights is 18 records = 18 KiB = 72
// registers -- small enough to be requested one LAYER ahead, while the team waits for the layer before.  Per
// layer a member then loads its input (each wave its own chunk of the whole board, into a wave-private LDS
// image: no workgroup barrier), runs 81 MFMAs per wave, adds the eight K parts through LDS, finishes three row
// fragments (bias, residual, ReLU, f16 hi/lo split) and hands its 3 KB of output to the team.
//
// Hand-off (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", measured form
// of its table's first row): every output byte is stored with an agent-scope relaxed atomic store (sc1,
// write-through), every storing wave drains its stores (vmcnt(0)), the workgroup meets at a barrier, ONE lane adds
// 1 to the team's counter (agent-scope atomic); a consumer's wave 0 polls that counter with agent-scope relaxed
// loads (sc1), the workgroup meets at a barrier, and every load of handed-off bytes is an agent-scope relaxed
// atomic load (sc1, never L1).  Counters are 64-bit and monotonic over the evaluator's lifetime (no reset between
// launches).  Every spin is bounded: a member that waits longer than ~1 s raises `status` (host-mapped) and the
// launch unwinds; the host turns that into an error at await.  Two team launches must not share a device at the
// same time (each would hold CUs the other's unscheduled members need): nsg_capi.hip keeps one token per device
// and a second evaluator falls back to the per-layer kernels.
//
// Arithmetic: kF16x3 (split f16 hi/lo, three f16 MFMAs per MAC, f32 accumulate) on the kF16x3 records and
// activation layout the evaluator keeps for batches without an MX plan -- a channel subset can be written
// without its neighbours (the MX formats share one exponent per 32 channels across FOUR members' outputs).
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

namespace nsg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

constexpr int kWaves = 8, kThreads = kWaves * 64;
constexpr int kEntries = 134;             // 24 + 110: one board with its halo (mfma_tile.h)
// one 16-byte piece of every entry; a multiple of 256 B: the four lane groups of a fragment read (four planes, the
// same entries) then fall on disjoint banks -- with 134 * 16 = 2144 B planes every read cost 12 LDS cycles instead of
// 4 and the eight waves' 54 reads per layer, not their 81 MFMAs, set the layer's compute time (5.0k of 15.6k cycles)
constexpr int kPlane = (kEntries * 16 + 255) / 256 * 256;
constexpr int kImage = 8 * kPlane;        // a wave's chunk: pieces 0-3 f16 hi, 4-7 f16 lo
constexpr int kLds = kWaves * kImage;     // 147 456 B: the K parts' accumulators alias the images' interior entries
static_assert(kLds <= 160 * 1024, "LDS");
constexpr int kItems = (81 * 8 + 63) / 64; // 16-byte items of a wave's chunk of the board: 81 rows x 128 B

// Pointers read out of the layer list are generic to the compiler (flat_ instructions); every one of them is global
// memory, and the hand-off form above is measured for global_ sc1 accesses only: cast before use.
#define NSG_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const NSG_GLOBAL T* asGlobal(const void* p) {
    return (const NSG_GLOBAL T*)(unsigned long long)p;
}
template <class T>
__device__ __forceinline__ NSG_GLOBAL T* asGlobal(void* p) {
    return (NSG_GLOBAL T*)(unsigned long long)p;
}
__device__ __forceinline__ u64 loadAgent(const void* p) { // global_load_dwordx2 ... sc1
    return __hip_atomic_load(asGlobal<u64>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void storeAgent(void* p, u64 v) { // global_store_dwordx2 ... sc1
    __hip_atomic_store(asGlobal<u64>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte sc1 load (there is no 16-byte atomic builtin).  The compiler does not count this load in vmcnt: the caller
// waits with waitLoads() before touching the result (its own waits only ever over-wait: vmcnt retires in order).
__device__ __forceinline__ u32x4 loadAgent16(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(asGlobal<u32x4>(p)) : "memory");
    return v;
}
__device__ __forceinline__ void waitLoads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int entryOf(int m) { // LDS entry of board row m (mfma_tile.h: 24 + (y+1)*10 + x)
    const int y = m / 9, x = m - y * 9;
    return 24 + (y + 1) * 10 + x;
}
__device__ __forceinline__ void splitPair(float a, float b, float floorV, unsigned& hi, unsigned& lo) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x2 x = {__builtin_amdgcn_fmed3f(a, floorV, 65000.f), __builtin_amdgcn_fmed3f(b, floorV, 65000.f)};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

#ifdef TEAM_STAMPS
__device__ u64 gTeamStamps[32 * 8];
#endif

// FR = row fragments per member: 3 (a board = 16 weight fragments x 2 row halves = 32 members, up to eight boards) or
// 1 (16 x 6 = 96 members per board, a single board -- the launch allows two --: a third of the MFMAs, fragment reads and tile rows per member,
// and the epilogue of one fragment; the members of a board then sit on every XCD, which costs a hand-off nothing
// measurable once the payload is stored sc1 -- it leaves the producer's L2 either way).
template <int FR>
__global__ __launch_bounds__(kThreads, 2) void teamTrunkKernel(const TeamLayer* __restrict__ layers, int nLayers,
                                                               int boards, u64* counters, TeamBases bases, int* status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kMembers = 16 * (6 / FR);
    constexpr int kSub = FR > 3 ? 3 : FR, kSubs = FR / kSub; // row fragments per fragment-read step, steps per tap
    const int team = FR == 3 ? (int)(blockIdx.x & 7) : (int)(blockIdx.x / kMembers);
    const int rank = FR == 3 ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x % kMembers);
    if (team >= boards || rank >= kMembers || team >= kTeamMaxBoards) return;
    const int j = rank & 15, h = rank >> 4; // weight fragment (16 output channels), row group (half or single fragment)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // = this wave's 32-channel chunk of K
    unsigned char* img = smem + wave * kImage;
    u64* ctr = counters + team * 8; // one 64-byte line per team
    const u64 base = bases.v[team];  // the counter's value when this launch's first layer starts

    // the image's halo entries stay zero for the whole launch: staging rewrites interior entries only
    for (int i = lane; i < kImage / 16; i += 64) reinterpret_cast<u32x4*>(img)[i] = u32x4{0u, 0u, 0u, 0u};

    int abase[FR]; // LDS read base of this lane's row in each of the member's row fragments
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int m = (h * FR + f) * 16 + li;
        abase[f] = ((m < 81 ? entryOf(m) : 11) - 11) * 16; // 11: every tap reads the zero entries
    }
    // staging: item k of a lane = piece lane / 8 of row k*8 + lane % 8 of this wave's chunk.  Eight consecutive
    // lanes write ONE piece of eight consecutive rows: consecutive entries of one plane, distinct banks (a lane per
    // piece of one row would put the eight pieces of an entry, 256-byte-aligned planes apart, on the same banks)
    // Only the board rows this member's fragments and their 3x3 neighbours touch are fetched: squares 0..57 for the
    // first row half (fragments 0-2 = squares 0..47, + their neighbours below), 38..80 for the second (63 % of the board on
    // average: a handed-off tile arrives at ~70 GB/s per CU, MI355X_MICROARCH.md handoff-payload)
    const int stPiece = lane >> 3;
    // (rows of the member's fragments -1 / +1 board row: 16*first - 10 .. 16*last + 25, clipped to the board)
    const int rowLo = (h * FR) * 16 - 10 > 0 ? (h * FR) * 16 - 10 : 0;
    const int rowHi = (h * FR + FR) * 16 + 9 < 80 ? (h * FR + FR) * 16 + 9 : 80;
    int srcRow[kItems], dstOff[kItems];
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int row = k * 8 + (lane & 7);
        srcRow[k] = (row >= rowLo && row <= rowHi) ? row : -1;
        dstOff[k] = stPiece * kPlane + entryOf(row < 81 ? row : 0) * 16;
    }
    // K parts: after its MFMAs a wave parks its three accumulator fragments in INTERIOR entries of its own image (the
    // halo entries must stay zero; the next layer's staging rewrites every interior entry it reads): 16-byte slot
    // f*64 + lane -> piece slot / (8 FR), square slot % (8 FR) -- squares inside the rows this member stages
    int redOff[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int slot = f * 64 + lane; // FR * 64 slots over the eight pieces: 8 * FR squares of each
        redOff[f] = (slot / (8 * FR)) * kPlane + entryOf(slot % (8 * FR)) * 16;
    }
    // output: lane (li, g) of fragment f holds channels ch0 .. ch0+3 of row (h*3+f)*16 + li
    const int ch0 = (j >> 2) * 64 + g * 16 + (j & 3) * 4;
    const int outOff = (ch0 >> 5) * 128 + (ch0 & 31) * 2; // byte offset of the hi pair inside a row; lo at +64

    auto loadWeights = [&](const TeamLayer& L, u32x4 (&w)[9][2]) {
        const int nkc = L.kdim / 32;
        const NSG_GLOBAL u32x4* wp = asGlobal<u32x4>(L.w) + ((size_t)wave * 18 * 16 + j) * 64 + lane; // record q = (chunk*9 + tap)*2 + s, 16 fragments
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                w[t][s] = wave < nkc ? wp[(size_t)(t * 2 + s) * 16 * 64] : u32x4{0u, 0u, 0u, 0u};
    };

#ifdef TEAM_STAMPS
#define TEAM_STAMP(I) if (blockIdx.x == 0 && tid == 0 && l >= 2 && l < 34) gTeamStamps[(l - 2) * 8 + (I)] = __builtin_amdgcn_s_memtime();
#else
#define TEAM_STAMP(I)
#endif
    u32x4 w[9][2];
    loadWeights(layers[0], w);
    bool alive = true;
    for (int l = 0; l < nLayers; ++l) {
        const TeamLayer L = layers[l];
        const int nkc = L.kdim / 32;
        const size_t inRow = (size_t)L.kdim * 4, outRow = (size_t)L.cout * 4;
        // ---- wait for the team's previous layer (layer 0 reads what the launch before this one wrote)
        // (the poller's first poll returns behind its own weight requests of a moment ago -- a wave's loads return in
        // order -- but the team needs that long to gather anyway: holding the poller's weights back until its poll had
        // matched cost 17-23 %, profiles/r03/g_team_trunk_ab.txt)
        TEAM_STAMP(0)
        if (l > 0) {
            if (tid == kThreads - 64) {
                const u64 target = base + (u64)l * kMembers;
                int spins = 0;
                while (loadAgent(ctr) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 4095) == 0 &&
                        (spins > (1 << 22) || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                        __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // bounded: ~1 s
                        alive = false;
                        break;
                    }
                }
            }
            alive = __syncthreads_and(alive ? 1 : 0) != 0; // (also: the polling wave's answer reaches every wave)
            if (!alive) return;
        }
        // ---- this wave's chunk of the board -> its LDS image (agent-scope loads: never the L1 of this CU)
        TEAM_STAMP(1)
        // (waves 0..2 also request their residual rows now: one round trip, hidden behind the tile and the MFMAs)
        const int mOut = (h * FR + (wave < FR ? wave : 0)) * 16 + li;
        const size_t rowOff = ((size_t)team * 81 + (mOut < 81 ? mOut : 0)) * outRow + outOff;
        u64 resHi = 0, resLo = 0;
        f32x4 biasV = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave < FR) {
            biasV = *asGlobal<f32x4>(L.bias + ch0);
            if (L.res && mOut < 81) {
                resHi = loadAgent(L.res + rowOff);
                resLo = loadAgent(L.res + rowOff + 64);
            }
        }
        if (wave < nkc) {
            u32x4 st[kItems];
            const unsigned char* xb = L.x + (size_t)team * 81 * inRow + (size_t)wave * 128 + (size_t)stPiece * 16;
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                st[k] = u32x4{0u, 0u, 0u, 0u};
                if (srcRow[k] >= 0) st[k] = loadAgent16(xb + (size_t)srcRow[k] * inRow);
            }
            waitLoads();
#pragma unroll
            for (int k = 0; k < kItems; ++k)
                if (srcRow[k] >= 0) *reinterpret_cast<u32x4*>(img + dstOff[k]) = st[k];
        }
        TEAM_STAMP(2)
        // (a wave reads only its own image: its own LDS writes are ordered before its reads, no barrier)
        f32x4 acc[FR];
#pragma unroll
        for (int f = 0; f < FR; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave < nkc) {
            // row fragments one tap ahead of their MFMAs (left to itself the compiler reads, waits, multiplies: the
            // layer's 54 LDS round trips in series were 5.0k of its 15.6k cycles)
            f16x8 xh[2][kSub], xl[2][kSub];
            auto readStep = [&](int st_, int buf) { // step = (tap, group of kSub fragments)
                const int t = st_ / kSubs, sub = st_ % kSubs;
                const int tapOff = ((t / 3 - 1) * 10 + (t % 3 - 1) + 11) * 16;
#pragma unroll
                for (int f = 0; f < kSub; ++f) {
                    xh[buf][f] = *reinterpret_cast<const f16x8*>(img + g * kPlane + abase[sub * kSub + f] + tapOff);
                    xl[buf][f] = *reinterpret_cast<const f16x8*>(img + (4 + g) * kPlane + abase[sub * kSub + f] + tapOff);
                }
            };
            readStep(0, 0);
#pragma unroll
            for (int st_ = 0; st_ < 9 * kSubs; ++st_) {
                if (st_ + 1 < 9 * kSubs) readStep(st_ + 1, (st_ + 1) & 1);
                const int t = st_ / kSubs, sub = st_ % kSubs;
                const f16x8 whi = __builtin_bit_cast(f16x8, w[t][0]), wlo = __builtin_bit_cast(f16x8, w[t][1]);
#pragma unroll
                for (int f = 0; f < kSub; ++f) {
                    f32x4& a = acc[sub * kSub + f];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xh[st_ & 1][f], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh[st_ & 1][f], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xl[st_ & 1][f], a, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0); // a step's reads stay in front of the step before's MFMAs
            }
        }
        TEAM_STAMP(3)
        // ---- add the K parts (fixed order: deterministic), waves 0..2 finish one row fragment each
#pragma unroll
        for (int f = 0; f < FR; ++f) *reinterpret_cast<f32x4*>(img + redOff[f]) = acc[f];
        __syncthreads();
        TEAM_STAMP(4)
        if (wave < FR) {
            const int ro = redOff[FR == 1 ? 0 : (wave < FR ? wave : 0)];
            f32x4 sum = *reinterpret_cast<const f32x4*>(smem + ro);
#pragma unroll
            for (int p = 1; p < kWaves; ++p) sum += *reinterpret_cast<const f32x4*>(smem + p * kImage + ro);
            if (mOut < 81) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(sum[r], L.accScale, biasV[r]);
                if (L.res) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    const f16x4 rh = __builtin_bit_cast(f16x4, resHi);
                    const f16x4 rl = __builtin_bit_cast(f16x4, resLo);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rh[r] + (float)rl[r];
                }
                const float floorV = L.relu ? 0.f : -65000.f;
                unsigned h01, l01, h23, l23;
                splitPair(v[0], v[1], floorV, h01, l01);
                splitPair(v[2], v[3], floorV, h23, l23);
                storeAgent(L.y + rowOff, (u64)h01 | ((u64)h23 << 32));
                storeAgent(L.y + rowOff + 64, (u64)l01 | ((u64)l23 << 32));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores have left before the barrier below
        }
        TEAM_STAMP(5)
        __syncthreads(); // every storing wave has drained, every parked accumulator has been read
        TEAM_STAMP(6)
        if (l + 1 < nLayers) {
            if (tid == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the next layer's weights go out now (read-only: plain loads) and are in flight while the team gathers;
            // issued BEHIND this wave's output stores and their drain, which would otherwise wait for them too.
            // (Two register sets, the next layer's records requested behind this layer's tile requests instead:
            // no faster -- a layer's 147 KB of records + 51 KB of tile are 3.1k cycles of this CU's L1 wherever
            // they are placed -- and the counted wait it needs is fragile.)
            loadWeights(layers[l + 1], w);
        }
    }
}

} // namespace

// Members per board: 96 (one row fragment each) for a single board, 32 (three each) for two to eight (measured,
// profiles/r03/g_team_trunk_members.txt: one board 4.76k against 3.34k evals/s, two boards 6.31k against 6.71k).
// NSG_TEAM_MEMBERS = 32 | 96 overrides (96: one or two boards only).
// Nine to sixteen boards: 16 members per board, one weight fragment x the whole board each.
int teamMembers(int boards) {
    static const int force = [] { const char* e = getenv("NSG_TEAM_MEMBERS"); return e ? atoi(e) : 0; }();
    if (boards > 8) return 16;
    if (force == 16) return 16;
    if (force == 32 || force == 96) return (force == 96 && boards > 2) ? 32 : force;
    return boards == 1 ? 96 : 32;
}

bool teamTrunkSupports(int channels, int stemKdim, int boards) {
    return channels == 256 && stemKdim % 32 == 0 && stemKdim <= 256 && boards >= 1 && boards <= kTeamMaxBoards;
}

#ifdef TEAM_STAMPS
void teamTrunkDumpStamps() {
    u64 h[32 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gTeamStamps), sizeof(h)) != hipSuccess) return;
    double d[7] = {0};
    for (int l = 0; l < 31; ++l) {
        for (int i = 0; i < 6; ++i) d[i] += (double)(h[l * 8 + i + 1] - h[l * 8 + i]);
        d[6] += (double)(h[(l + 1) * 8] - h[l * 8 + 6]);
    }
    fprintf(stderr, "team stamps (cycles/layer, member 0 lane 0): wait %.0f  issue-tile %.0f  mfma %.0f  red-write %.0f  reduce+epilogue %.0f  barrier %.0f  arrive+weights %.0f\n",
            d[0] / 31, d[1] / 31, d[2] / 31, d[3] / 31, d[4] / 31, d[5] / 31, d[6] / 31);
}
#endif

hipError_t launchTeamTrunk(const TeamLayer* devLayers, int nLayers, int boards, unsigned long long* counters,
                           const TeamBases& bases, int* status, hipStream_t stream) {
    if (nLayers < 1 || boards < 1 || boards > kTeamMaxBoards) return hipErrorInvalidValue;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)teamTrunkKernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)teamTrunkKernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)teamTrunkKernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (teamMembers(boards) == 96)
        hipLaunchKernelGGL(teamTrunkKernel<1>, dim3(boards * 96), dim3(kThreads), kLds, stream, devLayers, nLayers, boards,
                           counters, bases, status);
    else if (teamMembers(boards) == 16)
        hipLaunchKernelGGL(teamTrunkKernel<6>, dim3(boards * 16), dim3(kThreads), kLds, stream, devLayers, nLayers, boards,
                           counters, bases, status);
    else
        hipLaunchKernelGGL(teamTrunkKernel<3>, dim3(8 * 32), dim3(kThreads), kLds, stream, devLayers, nLayers, boards,
                           counters, bases, status);
    return hipGetLastError();
}

} // namespace nsg
