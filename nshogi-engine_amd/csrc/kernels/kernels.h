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

enum Precision { kFp32 = 0, kFp16 = 1, kBf16 = 2 };

inline int elemSize(int prec) { return prec == kFp32 ? 4 : 2; }
// Channels per 128-byte K chunk of the trunk convolution.
inline int chunkChannels(int prec) { return 128 / elemSize(prec); }

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
};
constexpr int kNfrag = 4; // every packed tensor uses 4 fragments (64 channels) per wave

// Picks a tile configuration for (batch, cout).
ConvPlan chooseConvPlan(int batch, int cout, int computeUnits);

// 3x3 convolution + folded-BN bias (+ residual) (+ ReLU) on
// activations [boards][81][cin] -> [boards][81][cout].  Buffers must hold
// paddedBoards(batch, plan.nb) boards.
inline int paddedBoards(int batch, int nb) { return (batch + nb - 1) / nb * nb; }
hipError_t launchConv3x3(const void* x, const void* wfrag, const float* bias,
                         const void* residual, void* y, int batch, int cin,
                         int cout, int relu, int prec, const ConvPlan& plan,
                         hipStream_t stream);

// Policy 1x1 conv (27 ch, +bias, raw logits -> policy[b][c*81+sq] f32) and
// value-feature 1x1 conv (VC ch, folded-BN bias, ReLU -> vfeat[b*vfeatStride + sq*VC+c] T)
// as ONE GEMM over coutPadded = roundup(VC+27, 64) channels ordered
// [value 0..VC) [policy VC..VC+27) [zero padding].
hipError_t launchHeads(const void* x, const void* wfrag, const float* bias,
                       float* policy, void* vfeat, int batch, int channels,
                       int coutPadded, int valueChannels, int vfeatStride,
                       int prec, hipStream_t stream);

// y[rows][cout] f32 = (ReLU)(x[rows][kdim] T * W + bias): value MLP layer 1.
hipError_t launchDense(const void* x, const void* wfrag, const float* bias,
                       float* y, int rows, int kdim, int cout, int relu,
                       int prec, hipStream_t stream);

// Value MLP layer 2 + squashing: o = w2 h + b2; value = (tanh(o0)+1)/2,
// draw = sigmoid(o1).  One wave per board, lane-shuffle reduction.
hipError_t launchValueOut(const float* h, const float* w2, const float* b2,
                          float* value, float* draw, int batch, int hidden,
                          hipStream_t stream);

// Fragment-ordered weights: number of 16-byte lane records INCLUDING the two
// trailing zero slabs the kernel's prefetch may touch.
size_t tileWeightRecords(int taps, int kdim, int cout, int prec);
// Host-side re-layout.  get(n, k, tap) returns the (already BN-folded) weight
// of output channel n, input channel k (< kReal), tap (< taps); input
// channels [kReal, kdim) are zero padding, as are output channels for which
// get() returns 0.
typedef float (*WeightGetter)(const void* ctx, int n, int k, int tap);
void packTileWeights(WeightGetter get, const void* ctx, int taps, int kReal,
                     int kdim, int cout, int prec, void* dst);

// Debug: activations [batch][81][c] T -> f32 [batch][c][81].
hipError_t launchActToNCHW(const void* x, float* dst, int batch, int c,
                           int prec, hipStream_t stream);

} // namespace nsg

#endif
