// kernels.h -- internal launch interface of the gfx950 kernels behind nsg.h.
//
// Device data layout (DESIGN.md "Data layout in HBM"):
//   bitboards   [B][C][2] u64            ml::FeatureBitboard, as the caller packed them
//   activations [Bpad][81][Cact] T       board-major, square, channel innermost ("NHWC")
//   weights     fragment-ordered 16-byte lane records (see conv3x3.hip)
//   policy      [B][27*81] f32           index c*81+sq, the reference's tensor (trt.cc:193-210)
//   value/draw  [B] f32
#ifndef NSG_KERNELS_H
#define NSG_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsg {

// kF16x3: every value is carried as an (f16 hi, f16 lo) pair, hi = f16(v),
// lo = f16(v - hi), and every product is evaluated as hi*hi + lo*hi + hi*lo on the
// f16 MFMA with f32 accumulation: ~22 significant bits (f32 has 24) at three
// 16-bit MFMAs per MAC instead of sixteen f32-MFMA cycles' worth.
//
// kF16m8 (trunk convolutions only): the main term w_hi*x_hi stays on the f16 MFMA, the
// two correction terms w_lo*x_hi and w_hi*x_lo -- each 2^-11 of the product -- are
// evaluated on fp8 (e4m3) copies of the operands by v_mfma_scale_f32_16x16x128_f8f6f4,
// which retires four times the K of the f16 instruction in twice its cycles: 2.1 instead
// of 3 MFMA units per MAC.  Activations: 128-byte chunks of 32 channels,
// [32 x f16 hi][32 x e4m3(hi)][32 x e4m3(lo * 2^12)].  The 1x1 heads and the value MLP of
// a kF16m8 evaluator run as kF16x3 (the last trunk layer writes the kF16x3 layout).
//
// kF16m6: kF16m8 with the correction operands in e2m3 (fp6) and ONE E8M0 scale per 32 input
// channels -- per (row, chunk) for the activation copies, per (output channel, tap, chunk) for
// the weight copies -- which the MX instruction applies itself.  With both operands in fp6 it
// retires its K = 128 in the cycles of ONE f16 MFMA (half the e4m3 form:
// profiles/r02/a_fp6_probe.txt): 1.5 MFMA units per MAC.  No fixed scales, hence no clamp
// window: the block exponent follows the data.  Activations: 128-byte chunks of 32 channels,
// [32 x f16 hi][24 B: e2m3(hi), channel order 0..31][1 B E8M0][3 B 0][1 B E8M0 again][3 B 0]
//              [24 B: e2m3(lo), channel order 0..31][1 B E8M0][3 B 0][1 B E8M0 again][3 B 0]
// (the exponent sits in both trailing dwords of a block: the conv's main loop reads the codes as 16 + 8 bytes and the
// exponent as the block's last dword, mfma_tile.h kSplitRead)
enum Precision { kFp32 = 0, kFp16 = 1, kBf16 = 2, kF16x3 = 3, kF16m8 = 4, kF16m6 = 5 };
// the two split-precision trunk formats that run their correction terms on the MX instruction
constexpr bool isMx(int prec) { return prec == kF16m8 || prec == kF16m6; }
constexpr int kM8LoShift = 12;   // stored lo byte = e4m3(lo * 2^12)
constexpr int kM8WLoShift = 10;  // weight record  = e4m3(w_lo * 2^10)
constexpr int kM8WHiShift = -2;  // weight record  = e4m3(w_hi * 2^-2); both products carry 2^-10
constexpr int kM8ScaleByte = 127 - 10; // E8M0 block scale of the weight operand

// Bytes one activation channel occupies (a kF16x3 pair is 2 + 2 bytes, kF16m8 2 + 1 + 1).
inline int elemSize(int prec) { return (prec == kFp32 || prec == kF16x3 || isMx(prec)) ? 4 : 2; }
// Precision of the 1x1 heads / value MLP of an evaluator whose trunk runs at `prec`.
inline int headPrecision(int prec) { return isMx(prec) ? (int)kF16x3 : prec; }
// Channels per 128-byte K chunk of the trunk convolution.
inline int chunkChannels(int prec) { return 128 / elemSize(prec); }
// MFMA K-slabs per (chunk, tap): two halves of the chunk, or for kF16x3 the three
// products (w_hi,x_hi) (w_lo,x_hi) (w_hi,x_lo) over the chunk's 32 channels.
inline int slabsPerTap(int prec) { return prec == kF16x3 ? 3 : 2; }
// Packed weight records per (chunk, tap): the two K halves, or for kF16x3 (w_hi, w_lo)
// -- w_hi serves both of its products from registers.
inline int recordsPerTap(int) { return 2; }
// 1-KiB-per-fragment weight records per channel chunk.  kF16m8 (chunks go in pairs A, B): per
// tap one f16 record for each chunk (w_hi) plus two records holding the 32 fp8 bytes per lane of
// the MX operand (k-group g: chunk g>>1, g&1 ? e4m3(w_hi) : e4m3(w_lo)) -- 4 per tap and pair.
inline int recordsPerChunk(int taps, int prec) {
    return isMx(prec) ? 2 * taps : taps * recordsPerTap(prec);
}
// Input channels are padded to whole chunks; kF16m8 to whole chunk pairs.
inline int inputChannelGranule(int prec) { return isMx(prec) ? 2 * chunkChannels(prec) : chunkChannels(prec); }

// ---- feature-plane expansion (reference K1/K2, src/cuda/extractbit.cu) ----
hipError_t launchExtractBitsNCHW(float* dst, const uint64_t* src, int batch,
                                 int channels, hipStream_t stream);
hipError_t launchExtractBitsNHWC(float* dst, const uint64_t* src, int batch,
                                 int channels, hipStream_t stream);
// Same bit selection, written straight into the trunk's activation layout
// [batch][81][cpad] in the trunk's element type (zero for c >= channels).
hipError_t launchExtractBitsAct(void* dst, const uint64_t* src, int batch,
                                int channels, int cpad, int prec,
                                hipStream_t stream);

// ---- MFMA tile kernels (mfma_tile.h) ----
struct ConvPlan {
    int nb;      // boards per workgroup
    int nfrag;   // 16-channel output fragments per wave (fixed at pack time)
    int nwaves;  // waves per workgroup
    int msplit = 1; // wave groups that split the tile's row fragments between them (small one-board tiles: 2)
    int ksplit = 1; // kF16m8 one-board tiles: waves of a channel group that split the input-channel chunk pairs between them
    int sslab = 1;  // MX two-board tiles at mid batches: waves of a channel group that split every chunk pair's slabs between them
};
constexpr int kNfrag = 4; // every packed tensor uses 4 fragments (64 channels) per wave

// Tuning overrides (0 = automatic), read from NSG_CONV_NB / _NWAVES / _NFRAG when an
// evaluator is created.
struct ConvTuning {
    int nb = 0, nwaves = 0, nfrag = 0, msplit = 0; // msplit: NSG_CONV_MSPLIT (1 = never split rows)
    int splitBatch = 1;    // NSG_SPLIT_BATCH=0: never run a batch as a full part + remainder
    int splitBatchMax3 = 9;  // NSG_SPLIT_BATCH_MAX: largest batch, in quarters of the CU count, that starts with a full chip of two-board tiles
    int rowsplit8Max = -1; // NSG_ROWSPLIT8_MAX_BATCH: largest batch whose four-way K split also splits the rows over two workgroups (-1: CUs / 8)
    int ksplit3 = 1;       // NSG_KSPLIT3=0: 192-channel nets keep the plans they had before the three-way K split
    int slabSplit = 0;     // NSG_SLAB_SPLIT=1: slab-split two-board tiles at mid batches (measured 8-11 % slower than the plans they would replace: opt-in)
    bool fullTilesOnly = false; // kF16m8: 4 fragments per wave at every batch size
};
ConvTuning readConvTuning();

// Picks a tile configuration for (batch, cout).
ConvPlan chooseConvPlan(int batch, int cout, int computeUnits, const ConvTuning& tune = ConvTuning());

// 3x3 convolution + folded-BN bias (+ residual) (+ ReLU) on
// activations [boards][81][cin] -> [boards][81][cout].  Buffers must hold
// paddedBoards(batch, plan.nb) boards.
inline int paddedBoards(int batch, int nb) { return (batch + nb - 1) / nb * nb; }
hipError_t launchConv3x3(const void* x, const void* wfrag, const float* bias,
                         const void* residual, void* y, int batch, int cin,
                         int cout, int relu, float accScale, int prec,
                         const ConvPlan& plan, hipStream_t stream,
                         unsigned long long* stamps = nullptr, bool outF16x3 = false);

// Persistent trunk: every 3x3 layer (stem + 2 per block) in ONE launch.  A layer
// list is built once with trunkLayersBytes()/fillTrunkLayer() on the host, uploaded,
// and reused for every batch size.  Needs cout == nwaves*64 (one workgroup covers all
// output channels of its boards); canRunTrunk() says whether a plan qualifies.
size_t trunkLayerBytes();
size_t trunkLayerStampsOffset(); // (diagnostic builds) byte offset of a layer's stamp pointer inside its list entry
void fillTrunkLayer(void* hostLayers, int index, const void* x, const void* wfrag,
                    const float* bias, const void* residual, void* y, int cin,
                    int cout, int relu, float accScale, bool outF16x3 = false);
bool canRunTrunk(int cout, const ConvPlan& plan);
hipError_t launchTrunk(const void* devLayers, int nLayers, int batch, int prec,
                       const ConvPlan& plan, hipStream_t stream);

// Cooperative trunk (mfma_tile.h, coopTrunkKernel): every 3x3 layer of a mid batch whose boards are shared by several
// workgroups (K-split plans) in ONE launch; workgroups of a board hand their output slices to each other through
// agent-scope stores / loads and one flag per (board, member) in `flags` (batch x members unsigned, zeroed before the
// launch).  kF16m6: 256 channels, the two-way K split (65 ... CUs/2 boards) and the four-way K split with its row
// groups (17 ... CUs/4); 192 channels, the three-way K split (17 ... CUs/3).  All batch x members workgroups must be
// resident at once.  `status`: host-mapped int raised when a bounded spin runs out.
bool canRunCoopTrunk(int cout, int stemKdim, int prec, const ConvPlan& plan, bool* firstSeparate = nullptr);
bool coopFits(int boards, int members, int computeUnits); // every workgroup of the launch resident at once (per XCD)
int coopMembers(int cout, const ConvPlan& plan); // workgroups per board
// (faultBoard >= 0: test hook -- that board's second member leaves at once, so its first waits in vain)
hipError_t launchCoopTrunk(const void* devLayers, int nLayers, int batch, int cout, int prec, const ConvPlan& plan,
                           unsigned* flags, unsigned flagBase, int* status, hipStream_t stream, int faultBoard);

// Team trunk (team_trunk.hip): every 3x3 layer of up to sixteen boards in ONE persistent launch, a board per team of
// 16 / 32 / 48 / 96 workgroups (12 / 24 / 36 / 72 for 192 trunk channels) that hand activations to each other through
// agent-scope stores / loads; the payload is its own flag (TeamHandoff).  kF16x3 arithmetic, records and activation
// layout; 256 or 192 trunk channels.  `status`: a host-mapped int the kernel raises when a bounded spin runs out.
typedef unsigned int team_u32x4 __attribute__((ext_vector_type(4)));
struct TeamLayer {
    const unsigned char* x;   // [boards][81][kdim] kF16x3
    const team_u32x4* w;      // kF16x3 records of the layer (packTileWeights)
    const float* bias;        // [cout]
    const unsigned char* res; // residual, layout of y, or null
    unsigned char* y;         // [boards][81][cout] kF16x3
    int kdim, cout, relu;
    float accScale;
};
constexpr int kTeamMaxBoards = 16;
// Hand-off buffers of a launch: `set` = four images of imageStride bytes (>= boards x 81 x 1024); layer l <
// nLayers - 1 writes image l % 4 of boards 0 .. boards-1.  When the launch starts every byte of those boards is 0xff;
// when it ends that holds again but for image (nLayers - 2) % 4.  `other` = the set the launch before this one used:
// the first cleanBoards boards of ITS image (nLayers - 2) % 4 are rewritten with 0xff.  Layer 0 reads TeamLayer::x,
// the last layer writes TeamLayer::y; TeamLayer::res != null means "the output of the layer two before" (a residual
// block's input).  nLayers >= 3.
struct TeamHandoff {
    unsigned char* set;
    unsigned char* other;
    size_t imageStride;
    int cleanBoards;
    // layer 0's input as the evaluator received it: [boards][bitChannels] 128-bit feature bitboards (the planes are
    // decoded while the first layer stages them: no plane buffer, no extraction launch); null: TeamLayer::x of layer 0
    const void* bits;
    int bitChannels;
};
bool teamTrunkSupports(int channels, int stemKdim, int boards);
// workgroups per board of a launch of `boards` boards of a `channels`-wide trunk on a device of `computeUnits` CUs
// (every member must be resident at once); 0: no team size fits.  forceRowGroups > 0: at most that many row groups.
int teamMembers(int boards, int channels, int computeUnits, int forceRowGroups);
#ifdef TEAM_STAMPS
void teamTrunkDumpStamps(); // diagnostic builds: per-phase cycles of one member, printed when an evaluator is destroyed
#endif
// (shortBy: test hook -- the grid is that many workgroups short, so the last team waits for members that never run)
hipError_t launchTeamTrunk(const TeamLayer* devLayers, int nLayers, int boards, int channels, int members,
                           const TeamHandoff& handoff, int* status, hipStream_t stream, int shortBy = 0);

// Policy 1x1 conv (27 ch, +bias, raw logits -> policy[b][c*81+sq] f32) and
// value-feature 1x1 conv (VC ch, folded-BN bias, ReLU -> vfeat[b*vfeatStride + sq*VC+c] T)
// as ONE GEMM over coutPadded = roundup(VC+27, 64) channels ordered
// [value 0..VC) [policy VC..VC+27) [zero padding].
hipError_t launchHeads(const void* x, const void* wfrag, const float* bias,
                       float* policy, void* vfeat, int batch, int channels,
                       int coutPadded, int valueChannels, int vfeatStride,
                       float accScale, int prec, hipStream_t stream);

// Value MLP layer 1, split over K: y[z][rows][cout] f32 = x[rows][K_z] T * W[K_z] for the
// denseSplits(kdim, prec) slices K_z of the input channels (a one-wave workgroup walks its
// chunks serially, one global-load latency each; 9 slices cut that chain 9x: 56 -> 9 us).
// The partial sums are added in a fixed order, with the bias and the ReLU, by launchValueOut.
int denseSplits(int kdim, int prec);
hipError_t launchDense(const void* x, const void* wfrag, const float* bias,
                       float* y, int rows, int kdim, int cout, size_t partStride,
                       float accScale, int prec, hipStream_t stream);

// Value MLP layer 1 epilogue (sum of the K slices + b1, ReLU) + layer 2 + squashing:
// o = w2 h + b2; value = (tanh(o0)+1)/2, draw = sigmoid(o1).  One wave per board.
hipError_t launchValueOut(const float* h, const float* b1, int nsplit, size_t partStride,
                          const float* w2, const float* b2,
                          float* value, float* draw, int batch, int hidden,
                          hipStream_t stream);

// Legal-move gather: out[offsets[b] + i] = policy[b][idx[offsets[b] + i]],
// optionally followed by a softmax over each position's moves (what the host does one leaf at
// a time in feedworker.cc:120-127 / frame.cc:105-118).  One wave per position.
hipError_t launchGatherLogits(const float* policy, const uint16_t* idx, const uint32_t* offsets,
                              float* out, int batch, int softmax, hipStream_t stream);

// Fragment-ordered weights: number of 16-byte lane records INCLUDING the eight
// trailing zero slabs the kernel's prefetch may touch.
size_t tileWeightRecords(int taps, int kdim, int cout, int prec);
// Host-side re-layout.  get(n, k, tap) returns the (already BN-folded) weight
// of output channel n, input channel k (< kReal), tap (< taps); input
// channels [kReal, kdim) are zero padding, as are output channels for which
// get() returns 0.
typedef float (*WeightGetter)(const void* ctx, int n, int k, int tap);
// `scale` multiplies every weight before conversion (a power of two chosen per
// tensor for kF16x3 so hi and lo stay in f16's normal range; the kernel undoes it
// exactly through accScale = 1/scale).
void packTileWeights(WeightGetter get, const void* ctx, int taps, int kReal,
                     int kdim, int cout, int prec, float scale, void* dst);

// Debug: activations [batch][81][c] T -> f32 [batch][c][81].
hipError_t launchActToNCHW(const void* x, float* dst, int batch, int c,
                           int prec, hipStream_t stream);

} // namespace nsg

#endif
