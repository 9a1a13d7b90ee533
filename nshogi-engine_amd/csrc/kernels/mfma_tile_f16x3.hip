// mfma_tile_f16x3.hip -- F16x3 (split f16 hi/lo, f32-equivalent) instantiations of
// the MFMA tile kernel (one translation unit per precision so the build parallelises).
#include "mfma_tile.h"

namespace nsg {
namespace tile {

hipError_t launchConvF16x3(const Args& a, int batch, const ConvPlan& p, hipStream_t s) {
    return launchConvPrec<kF16x3>(a, batch, p, s);
}
hipError_t launchTrunkF16x3(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s) {
    return launchTrunkPrec<kF16x3>(layers, n, batch, p, s);
}
// The heads of the smallest batches (the team trunk's, and the K-split plans'): with a handful of workgroups on the chip
// a one-wave workgroup streams its 64 output channels' records alone, a latency chain of ~10 us for the policy / value
// convolutions of ONE board; four waves of one fragment (16 channels) each run the four chains side by side.
// Threshold = rows of NSG_HEADS_SMALL_BOARDS boards (default 128: measured +3 % at one board, +1.7 % at 64, +0.4 % at 128;
// profiles/r04/zk_*).
static int smallHeadsBoards() {
    static const int v = [] { const char* e = getenv("NSG_HEADS_SMALL_BOARDS"); return e ? atoi(e) : 128; }();
    return v;
}
hipError_t launchHeadsF16x3(const Args& a, hipStream_t s) {
    if (a.totalRows <= smallHeadsBoards() * 81) {
        constexpr int kMF = 2;
        return launchOne<kF16x3, kHeads, kMF, 1, 4>(a, (a.totalRows + kMF * 16 - 1) / (kMF * 16), s);
    }
    return launchHeadsPrec<kF16x3>(a, s);
}
hipError_t launchDenseF16x3(const Args& a, hipStream_t s) {
    if (a.totalRows <= smallHeadsBoards()) {
        constexpr int kMF = 2;
        return launchOne<kF16x3, kDense, kMF, 1, 4>(a, (a.totalRows + kMF * 16 - 1) / (kMF * 16), s);
    }
    return launchDensePrec<kF16x3>(a, s);
}

} // namespace tile
} // namespace nsg
