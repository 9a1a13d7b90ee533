// team_trunk.hip -- the whole 3x3 trunk of up to sixteen boards in ONE persistent launch, for the smallest
// batches (the engine's leaf batches at the start of a search, batch-1 analysis).
//
// Why: at one board a 3x3 layer is 2.36 MB of weights against 15 MFLOP -- a weight-bandwidth problem.  The
// per-layer tile kernels give a board to 24 workgroups that each stream 622 KB of records through ONE L1 for 108
// MFMAs per wave, behind a launch boundary, a kernel-start latency and a tile round trip per layer: 14.5 us per
// layer.  Here a board belongs to a TEAM of 96 / 48 / 32 / 16 workgroups (teamMembers() below; one per CU).  Member
// (j, h) computes 16 output channels (weight fragment j) of row group h -- 1, 2, 3 or all 6 of the board's 16-row
// fragments: its eight waves split K by 32-channel chunk, so a wave's share of a layer's weights is 18 records =
// 18 KiB = 72 registers, requested as soon as the layer before has stored its output.  Per layer a member loads
// its input (each wave its own chunk of the rows it needs, into a wave-private LDS image: no workgroup barrier),
// runs 27 MFMAs per wave and row fragment, adds the eight K parts through LDS, finishes its row fragments (bias,
// residual, ReLU, f16 hi/lo split) and hands its output to the team.  (Block b computes weight fragment b % 16 -- every
// team size is a multiple of 16 -- so under the observed round-robin placement an XCD's L2 holds the records of two
// fragments for every board: one copy of the records leaves memory per launch.  Teams that each lived on one XCD
// fetched them eight times over, 774 MB per launch at eight boards: profiles/r03/README.md.  Speed only.)
//
// Hand-off: the payload is its own flag.  The layers but the last write into four rotating images (`TeamHandoff::
// set`, [boards][81 rows][1024 B]; layer l writes image l % 4) that hold the SENTINEL -- all bits set -- where
// nothing has been handed over yet.  A producing lane stores its piece -- four channels' f16 hi halves and lo halves,
// 16 bytes -- with one agent-scope store (sc1, write-through) and does not wait for it.  A consumer requests its
// input tile with 16-byte agent-scope loads (sc1, never this CU's L1), compares every word of a piece with the
// sentinel and requests the pieces that were not there yet again, until all are.  No word a layer stores is the
// sentinel: a stored half is an f16 of a number clamped to +-65000, or of the difference between such a number and
// its f16 rounding; 0xffff is a NaN.  (So a piece need not even arrive in one piece.)  A layer therefore costs ONE
// one-way trip of its payload on the critical path; the counter protocol this replaced paid for a store drain, an
// atomic add, a poll and then the tile's round trip (profiles/r03/README.md).
//   The images are this kernel's own (layer 0 reads, the last layer writes the evaluator's kF16x3 layout): a row is
// [chunk][member % 4][lane group % 2] pieces, so that a wave's chunk of a row is 128 contiguous bytes and a producing
// wave's store touches 2 x 32 bytes of a row.  In the evaluator's layout the same store is 128 scattered 8-byte
// pieces per wave, and a write-through store is paid per segment it touches: +12 % at 3-16 boards.
//   Reuse of an image.  Write C(n, l) for the moment ALL EIGHT waves of member n have their input of layer l (the
// workgroup barrier behind the MFMAs; the input includes the output of layer l - 1 of every neighbour -- every member
// whose rows n reads, which are the members that read n's) and F(n, l) > C(n, l) for n's output stores of layer l.
// Behind C(n, l) member n puts the sentinel back over its own output of layer l - 2 (the residual of layer l,
// requested before the tile, has arrived by then): every neighbour m has stored layer l - 1, so has read layer
// l - 2.  The storing waves wait for all their memory operations (vmcnt(0)) before C(n, l + 1) < F(n, l + 1).  A member
// m that looks at that place again -- for layer l + 2's output, after F(m, l + 2) > C(m, l + 2) > F(n, l + 1) -- finds
// the sentinel or the new value, never the old one; and n overwrites layer l - 2's image with layer l + 2's after
// C(n, l + 2) > F(m, l + 1) > C(m, l + 1): after every neighbour has read layer l.
//   Across launches.  The last layer but one's image is still being read when the launch ends: it is left as it is.
// Two sets alternate between launches and a launch restores the sentinel in that one image of the OTHER set
// (`cleanBoards` boards of it; the kernel boundary orders that against the set's next use).
//   Every access to an image is agent-scope (sc1), the sentinel stores too: a plain store's line can sit in one
// XCD's L2 while another XCD's sc1 load reads the stale piece behind it from memory -- and takes it for a new one.
//   Every spin is bounded: a member that waits longer than ~1 s raises `status` (host-mapped) and the launch unwinds;
// the host turns that into an error at await.  Two team launches must not share a device at the same time (each
// would hold CUs the other's unscheduled members need): nsg_capi.hip keeps one token per device.
//
// Arithmetic: kF16x3 (split f16 hi/lo, three f16 MFMAs per MAC, f32 accumulate) on the kF16x3 records the evaluator
// keeps for batches without an MX plan -- a channel subset can be written without its neighbours (the MX formats
// share one exponent per 32 channels across FOUR members' outputs).
#include "bitboard.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <atomic>
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
// piece p of an image starts (p & 3) * kSkew bytes into its plane: a staging store writes, per group of eight lanes,
// the four pieces of two neighbouring rows (the lane order that makes the global load of a row's pieces one line per
// lane quad) -- same entry, four planes a multiple of 128 bytes apart, would be a 4-way bank conflict on every store
constexpr int kSkew = 32;
static_assert(kEntries * 16 + 3 * kSkew <= kPlane, "the skewed pieces must fit their planes");
constexpr int kImage = 8 * kPlane;        // a wave's chunk: pieces 0-3 f16 hi, 4-7 f16 lo
constexpr int kFlagOff = kWaves * kImage; // one int behind the images: raised by a wave whose bounded wait ran out
constexpr int kLds = kFlagOff + 16;       // 147 472 B: the K parts' accumulators alias the images' interior entries
static_assert(kLds <= 160 * 1024, "LDS");

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
// 16-byte agent-scope accesses (there is no 16-byte atomic builtin): raw buffer loads / stores with the sc1 bit
// (buffer_load_dwordx4 ... offen sc1).  Compiler-visible on purpose: a first version issued global_load_dwordx4 sc1
// from an asm statement and waited in another, and the scheduler moved register-only readers of the loaded values (a
// v_max3 of a piece's words) in front of the wait -- nothing orders an asm's outputs against a later asm.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t bufferOf(const void* p) { // byte-addressed, no bounds to speak of
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00027000);
}
__device__ __forceinline__ u32x4 loadAgent16(rsrc_t r, int byteOff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byteOff, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void storeAgent16(rsrc_t r, int byteOff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byteOff, 0, 16 /* sc1 */);
}
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) {
    const unsigned ab = a > b ? a : b;
    return ab > c ? ab : c; // v_max3_u32
}
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

// FR = row fragments per member (1, 2, 3 or all 6: 96, 48, 32 or 16 members per board).
// nj = weight fragments of a layer = trunk channels / 16 (16 for 256 channels, 12 for 192): a team has nj members per
// row group.
template <int FR>
__global__ __launch_bounds__(kThreads, 2) void teamTrunkKernel(const TeamLayer* __restrict__ layers, int nLayers,
                                                               int boards, int nj, TeamHandoff ho, int* status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int kMembers = nj * (6 / FR);
    constexpr int kSub = FR > 3 ? 3 : FR, kSubs = FR / kSub; // row fragments per fragment-read step, steps per tap
    // (kMembers is a multiple of nj: block b computes weight fragment b % nj, and under the observed round-robin
    // placement XCD b % 8 -- an XCD's L2 holds the records of two (nj = 16) or three (nj = 12) fragments, for every board)
    const int team = (int)(blockIdx.x / kMembers);
    const int rank = (int)(blockIdx.x % kMembers);
    if (team >= boards || rank >= kMembers || team >= kTeamMaxBoards) return;
    const int j = rank % nj, h = rank / nj; // weight fragment (16 output channels), row group (half or single fragment)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // = this wave's 32-channel chunk of K
    unsigned char* img = smem + wave * kImage;
    const int member = team * kMembers + rank, nActive = boards * kMembers; // (cleaning shares)

    int* gaveUp = reinterpret_cast<int*>(smem + kFlagOff);
    if (tid == 0) *gaveUp = 0;
    // the image's halo entries stay zero for the whole launch: staging rewrites interior entries only
    for (int i = lane; i < kImage / 16; i += 64) reinterpret_cast<u32x4*>(img)[i] = u32x4{0u, 0u, 0u, 0u};

    int abase[FR]; // LDS read base of this lane's row in each of the member's row fragments
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int m = (h * FR + f) * 16 + li;
        abase[f] = ((m < 81 ? entryOf(m) : 11) - 11) * 16 + g * kSkew; // 11: every tap reads the zero entries; pieces g and 4 + g
    }
    // staging.  Only the board rows this member's fragments and their 3x3 neighbours touch are fetched: squares 0..57
    // for the first row half of a 32-member team (fragments 0-2 = squares 0..47, + their neighbours below), 38..80
    // for the second
    // (rows of the member's fragments -1 / +1 board row: 16*first - 10 .. 16*last + 25, clipped to the board)
    const int rowLo = (h * FR) * 16 - 10 > 0 ? (h * FR) * 16 - 10 : 0;
    const int rowHi = (h * FR + FR) * 16 + 9 < 80 ? (h * FR + FR) * 16 + 9 : 80;
    auto entry16 = [](int row) { return (34 + row + ((row * 57) >> 9)) * 16; }; // entryOf(row) * 16, row < 128
    // K parts: after its MFMAs a wave parks its three accumulator fragments in INTERIOR entries of its own image (the
    // halo entries must stay zero; the next layer's staging rewrites every interior entry it reads): 16-byte slot
    // f*64 + lane -> piece slot / (8 FR), square slot % (8 FR) -- squares inside the rows this member stages
    int redOff[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        const int slot = f * 64 + lane; // FR * 64 slots over the eight pieces: 8 * FR squares of each
        redOff[f] = (slot / (8 * FR)) * kPlane + ((slot / (8 * FR)) & 3) * kSkew + entryOf(slot % (8 * FR)) * 16;
    }
    // output: lane (li, g) of fragment f holds channels ch0 .. ch0+3 of row (h*3+f)*16 + li
    const int ch0 = (j >> 2) * 64 + g * 16 + (j & 3) * 4;
    const int outOff = (ch0 >> 5) * 128 + (ch0 & 31) * 2; // byte offset of the hi pair inside a row; lo at +64
    // ... in the evaluator's kF16x3 layout (layer 0's input, the last layer's output).  The hand-off images are this
    // kernel's own.  A lane's piece of an image row = its four channels' hi halves + lo halves, 16 bytes; a row =
    // [chunk kc][member j & 3][g & 1]: the pieces of one 32-channel chunk (which come from members 4 (kc / 2) .. + 3,
    // lane groups g = 2 (kc & 1), + 1) are 128 contiguous bytes for the wave that reads that chunk, and a producing
    // wave's store touches 2 x 32 bytes per row -- 32 line segments, not the 128 scattered 8-byte pieces of the
    // evaluator's layout (write-through: a store is paid per segment; profiles/r03/README.md).
    const int imgOff = (2 * (j >> 2) + (g >> 1)) * 128 + (j & 3) * 32 + (g & 1) * 16;
    // staging from an image: a lane takes BOTH pieces (members jq = 2 jp, 2 jp + 1; 32 bytes apart) of group gq of
    // row 16 k + lane / 4: channels 16 gq + 8 jp .. + 7 of the chunk = LDS piece 2 gq + jp whole (hi; + 4 lo), one
    // 16-byte LDS write each.  The four lanes of a QUAD take the four (gq, jp) of ONE row: each of their two loads
    // touches one 128-byte line (with the row in lane % 16 a quad read four rows, four lines, and the texture unit
    // took ~3x as long per wave load: mfma_tile.h, staging); eight consecutive lanes write two rows x four pieces,
    // on eight different 16-byte slots thanks to the pieces' skew.
    const int stGq = lane & 1, stJp = (lane >> 1) & 1, sli = lane >> 2;
    const int imgSrc = (2 * stJp) * 32 + stGq * 16; // inside the wave's chunk of a row
    const int imgDst = (2 * stGq + stJp) * (kPlane + kSkew);
    // items (16 rows each) a member stages: the rows of its fragments and one board row either side = at most FR + 2
    // consecutive items from item h * FR - 1 on
    constexpr int kImgItems = FR + 2 < 6 ? FR + 2 : 6;
    const int item0 = h * FR - 1 > 0 ? (h * FR - 1 < 6 - kImgItems ? h * FR - 1 : 6 - kImgItems) : 0;

    auto loadWeights = [&](const TeamLayer& L, u32x4 (&w)[9][2]) {
        const int nkc = L.kdim / 32;
        const NSG_GLOBAL u32x4* wp = asGlobal<u32x4>(L.w) + ((size_t)wave * 18 * nj + j) * 64 + lane; // record q = (chunk*9 + tap)*2 + s, nj fragments
        if (wave < nkc) { // (uniform: a branch, not a select that would need the loaded value at once)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) w[t][s] = wp[(size_t)(t * 2 + s) * nj * 64];
        }
    };

    // Diagnostic build -DTEAM_JITTER (never shipped): every member idles a pseudo-random time (up to ~30 us, longer
    // than a layer) at the points that matter for the hand-off argument at the head of the file -- before its tile
    // requests, before its output stores, before it restores the sentinel -- so that members drift layers apart as
    // far as the protocol lets them.  tests/test_gpu_evaluator.py::test_team_trunk_handoff_images_across_launches
    // passes on that build (profiles/r03/README.md).
#ifdef TEAM_JITTER
#define TEAM_IDLE(POINT)                                                                                        \
    {                                                                                                           \
        unsigned hsh = (blockIdx.x * 2654435761u) ^ ((unsigned)l * 40503u) ^ ((POINT) * 2246822519u);            \
        hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;                                                 \
        if ((hsh & 3u) == 0)                                                                                    \
            for (unsigned i = 0; i < (hsh >> 27); ++i) __builtin_amdgcn_s_sleep(32);                            \
    }
#else
#define TEAM_IDLE(POINT)
#endif
#ifdef TEAM_STAMPS
#define TEAM_STAMP(I) if (blockIdx.x == 0 && tid == 0 && l >= 2 && l < 34) gTeamStamps[(l - 2) * 8 + (I)] = __builtin_amdgcn_s_memtime();
#else
#define TEAM_STAMP(I)
#endif
    // (A second record set for the 96-member teams, filled behind a layer's first round of tile requests so that the
    // next layer's tile does not wait behind its 147 KB of records: measured 15 % slower, one board 4.87k against 5.70k
    // evals/s -- the records then sit in front of the layer's output stores.  profiles/r03/README.md)
    u32x4 w[9][2];
    loadWeights(layers[0], w);
    for (int l = 0; l < nLayers; ++l) {
        const TeamLayer L = layers[l];
        const int nkc = L.kdim / 32;
        const size_t inRow = (size_t)L.kdim * 4, outRow = (size_t)L.cout * 4;
        // layer l reads image (l - 1) % 4 of the set and writes image l % 4; the last layer writes the evaluator's
        const unsigned char* xPtr = l == 0 ? L.x : ho.set + (size_t)((l - 1) & 3) * ho.imageStride;
        unsigned char* yPtr = l + 1 == nLayers ? L.y : ho.set + (size_t)(l & 3) * ho.imageStride;
        unsigned char* oldPtr = l >= 2 ? ho.set + (size_t)((l - 2) & 3) * ho.imageStride : nullptr; // this member's output of layer l - 2
        const unsigned char* resPtr = L.res ? oldPtr : nullptr;
        const bool fromImage = l > 0, toImage = l + 1 < nLayers;
        TEAM_STAMP(0)
        // (waves 0..FR-1 also request their residual rows now -- their own stores of two layers ago, long landed)
        const int mOut = (h * FR + (wave < FR ? wave : 0)) * 16 + li;
        const int rowBase = (team * 81 + (mOut < 81 ? mOut : 0)) * (int)outRow; // (an image row holds the trunk's channels too: the same row size)
        const int rowImg = rowBase + imgOff;
        const rsrc_t xBuf = bufferOf(xPtr), yBuf = bufferOf(yPtr), oldBuf = bufferOf(oldPtr ? oldPtr : yPtr);
        u32x4 resV = u32x4{0u, 0u, 0u, 0u};
        f32x4 biasV = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave < FR) {
            biasV = *asGlobal<f32x4>(L.bias + ch0);
            if (resPtr && mOut < 81) resV = loadAgent16(oldBuf, rowImg);
        }
        TEAM_STAMP(1)
        TEAM_IDLE(1)
        // ---- this wave's chunk of the board -> its LDS image; a piece is there when neither of its halves is the
        // sentinel (layer 0 reads the planes of the launch before this one: there at the first request)
        if (wave < nkc && !fromImage && ho.bits) {
            // layer 0 straight from the feature bitboards (bitboard.h): this lane's LDS piece = channels 32 wave +
            // 8 (2 gq + jp) .. + 7 of its rows; the f16 hi / lo split of extractAct<kF16x3>, bit for bit
            typedef unsigned long long u64_;
            const int c0 = wave * 32 + (2 * stGq + stJp) * 8;
            u32x4 bb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                bb[i] = c0 + i < ho.bitChannels
                            ? asGlobal<u32x4>(ho.bits)[(size_t)team * ho.bitChannels + c0 + i]
                            : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < kImgItems; ++k) {
                const int row = (item0 + k) * 16 + sli;
                if (row < rowLo || row > rowHi) continue;
                unsigned hi[4], lo[4];
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const float a = __uint_as_float(selectBit(((u64_)bb[i].y << 32) | bb[i].x, ((u64_)bb[i].w << 32) | bb[i].z, row));
                    const float b = __uint_as_float(selectBit(((u64_)bb[i + 1].y << 32) | bb[i + 1].x,
                                                              ((u64_)bb[i + 1].w << 32) | bb[i + 1].z, row));
                    splitPair(a, b, -65000.f, hi[i / 2], lo[i / 2]);
                }
                unsigned char* d = img + imgDst + entry16(row);
                *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4*>(d + 4 * kPlane) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            }
        } else if (wave < nkc) {
            // item k of a lane = two 16-byte pieces of row 16 k + lane % 16 that make LDS piece 2 gq + jp (hi) and
            // its lo twin: from an image the two members' [hi4 | lo4] (32 bytes apart), from the evaluator's layout
            // (layer 0) the hi piece and the lo piece themselves (64 bytes apart)
            u32x4 st[2 * kImgItems];
            const int xb = team * 81 * (int)inRow + wave * 128 + (fromImage ? imgSrc : (2 * stGq + stJp) * 16);
            const int second = fromImage ? 32 : 64;
            unsigned pend = 0;
#pragma unroll
            for (int k = 0; k < kImgItems; ++k) {
                const int row = (item0 + k) * 16 + sli;
                if (row >= rowLo && row <= rowHi) pend |= 1u << k;
            }
            auto request = [&] {
                asm volatile("" ::: "memory"); // every round asks memory again
#pragma unroll
                for (int k = 0; k < kImgItems; ++k)
                    if ((pend >> k) & 1u) {
                        const int src = xb + ((item0 + k) * 16 + sli) * (int)inRow;
                        st[2 * k] = loadAgent16(xBuf, src);
                        st[2 * k + 1] = loadAgent16(xBuf, src + second);
                    }
            };
            auto take = [&] {
#pragma unroll
                for (int k = 0; k < kImgItems; ++k) {
                    if (!((pend >> k) & 1u)) continue;
                    const u32x4 a = st[2 * k], b = st[2 * k + 1];
                    // (no word of a stored piece is all ones: the largest of the eight tells)
                    const unsigned top = umax3(umax3(a.x, a.y, a.z), umax3(a.w, b.x, b.y), b.z > b.w ? b.z : b.w);
                    if (top != 0xffffffffu) {
                        unsigned char* d = img + imgDst + entry16((item0 + k) * 16 + sli);
                        *reinterpret_cast<u32x4*>(d) = fromImage ? u32x4{a.x, a.y, b.x, b.y} : a;
                        *reinterpret_cast<u32x4*>(d + 4 * kPlane) = fromImage ? u32x4{a.z, a.w, b.z, b.w} : b;
                        pend &= ~(1u << k);
                    }
                }
            };
            int spins = 0;
            request();
            take();
            while (__builtin_amdgcn_ballot_w64(pend != 0) != 0) {
                if ((++spins & 255) == 0 &&
                    (spins > (1 << 20) || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                    __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // bounded: ~1 s
                    *gaveUp = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                request();
                take();
            }
#ifdef TEAM_STAMPS
            if (blockIdx.x == 0 && lane == 0 && l >= 2 && l < 34) atomicAdd(&gTeamStamps[(l - 2) * 8 + 7], (u64)spins + 1);
#endif
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
        // ---- add the K parts (fixed order: deterministic), waves 0..FR-1 finish one row fragment each
#pragma unroll
        for (int f = 0; f < FR; ++f) *reinterpret_cast<f32x4*>(img + redOff[f]) = acc[f];
        // (the sentinel stores of the layer before have landed: the argument at the head of the file wants that
        // before C(n, l); they were issued in front of this layer's tile requests, whose data the MFMAs above used)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (*gaveUp) return; // (a wave whose wait ran out reaches every wave of the member; the others see `status`)
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
                if (resPtr) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    const f16x4 rh = __builtin_bit_cast(f16x4, (u64)resV.x | ((u64)resV.y << 32));
                    const f16x4 rl = __builtin_bit_cast(f16x4, (u64)resV.z | ((u64)resV.w << 32));
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rh[r] + (float)rl[r];
                }
                TEAM_IDLE(2)
                const float floorV = L.relu ? 0.f : -65000.f;
                unsigned h01, l01, h23, l23;
                splitPair(v[0], v[1], floorV, h01, l01);
                splitPair(v[2], v[3], floorV, h23, l23);
                if (toImage) {
                    storeAgent16(yBuf, rowImg, u32x4{h01, h23, l01, l23});
                } else {
                    storeAgent(yPtr + (size_t)(rowBase + outOff), (u64)h01 | ((u64)h23 << 32));
                    storeAgent(yPtr + (size_t)(rowBase + outOff) + 64, (u64)l01 | ((u64)l23 << 32));
                }
                // behind the barrier above = C(n, l), every wave of the member has its input: the sentinel goes
                // back over this member's output of layer l - 2 (see the head of the file)
                TEAM_IDLE(3)
                if (oldPtr) storeAgent16(oldBuf, rowImg, u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
            }
        }
        TEAM_STAMP(5)
        __syncthreads(); // every parked accumulator has been read: the next layer's staging may overwrite them
        TEAM_STAMP(6)
        // ---- the next layer's records go out BEHIND the member's output stores (requested in front of them -- by
        // any wave of the member: a workgroup has one address path -- they held the stores, and with them the team,
        // back by the time 147 KB of records take to arrive: profiles/r03/README.md).
        if (l + 1 < nLayers) loadWeights(layers[l + 1], w);
        // the image the launch before this one left behind in the other set gets its sentinel back
        if (l == 0) {
            const size_t nPieces = (size_t)ho.cleanBoards * (81 * 1024 / 16);
            // (agent-scope stores like every other access to an image: plain stores here were not seen by the
            // agent-scope loads of the launch after the next on another XCD -- stale pieces, taken for new ones)
            const rsrc_t other = bufferOf(ho.other + (size_t)((nLayers - 2) & 3) * ho.imageStride);
            for (size_t i = (size_t)member * kThreads + tid; i < nPieces; i += (size_t)nActive * kThreads)
                storeAgent16(other, (int)(i * 16), u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
        }
    }
}

} // namespace

// Members per board = weight fragments (trunk channels / 16: 16 or 12) x row groups, as many as fit the chip's CUs (a
// member owns a CU: 147 KB of LDS): with 16 fragments on 256 CUs 96 (one row fragment each) for one or two boards,
// 48 (two) for three to five, 32 (three) for six to eight, 16 (the whole board) for nine to sixteen.  Measured, evals/s
// at 2 / 3 / 4 / 5 boards: 96 members 11.2k / - / - / -, 48 members 9.2k / 13.5k / 17.7k / 21.6k, 32 members 7.8k /
// 11.5k / 15.4k / 18.8k; 6 / 8 boards: 32 members 22.5k / 29.3k, 16 members 16.2k / 21.3k
// (profiles/r03/g_team_trunk_members.txt).  Every member must be resident at once (they wait for each other): a
// device with fewer CUs (a partition, a CU mask) gets the largest team that fits, or none -- 0 -- and the caller
// runs the per-layer kernels.  `forceRowGroups` (NSG_TEAM_MEMBERS / 16) asks for fewer row groups than that.
int teamMembers(int boards, int channels, int computeUnits, int forceRowGroups) {
    const int nj = channels / 16;
    for (int rg : {6, 3, 2, 1}) {
        if (forceRowGroups > 0 && rg > forceRowGroups) continue;
        if ((long)boards * nj * rg <= computeUnits) return nj * rg;
    }
    return 0;
}

// 256 or 192 trunk channels: a member's eight waves take one 32-channel chunk of K each (a wave's share of a layer's
// records is 72 registers), so at most eight chunks; rows of an image are 1024 bytes.
bool teamTrunkSupports(int channels, int stemKdim, int boards) {
    return (channels == 256 || channels == 192) && stemKdim % 32 == 0 && stemKdim <= 256 && boards >= 1 &&
           boards <= kTeamMaxBoards;
}

#ifdef TEAM_STAMPS
void teamTrunkDumpStamps() {
    u64 h[32 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(gTeamStamps), sizeof(h)) != hipSuccess) return;
    double d[7] = {0}, tries = 0;
    for (int l = 0; l < 31; ++l) tries += (double)h[l * 8 + 7];
    fprintf(stderr, "team stamps: %.2f tile request rounds per layer, the member's eight waves together\n", tries / 31);
    u64 zero[32 * 8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gTeamStamps), zero, sizeof(zero));
    for (int l = 0; l < 31; ++l) {
        for (int i = 0; i < 6; ++i) d[i] += (double)(h[l * 8 + i + 1] - h[l * 8 + i]);
        d[6] += (double)(h[(l + 1) * 8] - h[l * 8 + 6]);
    }
    fprintf(stderr, "team stamps (cycles/layer, member 0 lane 0): residual request %.0f  tile (wait + stage) %.0f  mfma %.0f  next weights + clean + park + barrier %.0f  reduce + epilogue %.0f  barrier %.0f  loop %.0f\n",
            d[0] / 31, d[1] / 31, d[2] / 31, d[3] / 31, d[4] / 31, d[5] / 31, d[6] / 31);
}
#endif

hipError_t launchTeamTrunk(const TeamLayer* devLayers, int nLayers, int boards, int channels, int members,
                           const TeamHandoff& handoff, int* status, hipStream_t stream, int shortBy) {
    const int nj = channels / 16;
    if (nLayers < 3 || boards < 1 || boards > kTeamMaxBoards || !handoff.set || !handoff.other ||
        !teamTrunkSupports(channels, 32, boards) || members % nj != 0 ||
        handoff.imageStride < (size_t)boards * 81 * 1024 || handoff.imageStride < (size_t)handoff.cleanBoards * 81 * 1024)
        return hipErrorInvalidValue;
    const int rowGroups = members / nj;
    if (rowGroups != 1 && rowGroups != 2 && rowGroups != 3 && rowGroups != 6) return hipErrorInvalidValue;
    // (the attribute belongs to a function ON a device: one process may drive several -- selfplay --num-gpus)
    static std::atomic<unsigned long long> attrDevMask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (!(attrDevMask.load() & (1ull << (dev & 63)))) {
        const void* kernels[] = {(const void*)teamTrunkKernel<1>, (const void*)teamTrunkKernel<2>,
                                 (const void*)teamTrunkKernel<3>, (const void*)teamTrunkKernel<6>};
        for (const void* k : kernels) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
            if (e != hipSuccess) return e;
        }
        attrDevMask.fetch_or(1ull << (dev & 63));
    }
    if (shortBy < 0 || shortBy >= boards * members) return hipErrorInvalidValue;
    const dim3 grid(boards * members - shortBy), block(kThreads);
    if (rowGroups == 6)
        hipLaunchKernelGGL(teamTrunkKernel<1>, grid, block, kLds, stream, devLayers, nLayers, boards, nj, handoff, status);
    else if (rowGroups == 3)
        hipLaunchKernelGGL(teamTrunkKernel<2>, grid, block, kLds, stream, devLayers, nLayers, boards, nj, handoff, status);
    else if (rowGroups == 1)
        hipLaunchKernelGGL(teamTrunkKernel<6>, grid, block, kLds, stream, devLayers, nLayers, boards, nj, handoff, status);
    else
        hipLaunchKernelGGL(teamTrunkKernel<3>, grid, block, kLds, stream, devLayers, nLayers, boards, nj, handoff, status);
    return hipGetLastError();
}

} // namespace nsg
