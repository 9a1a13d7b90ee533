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
hipError_t launchHeadsF16x3(const Args& a, hipStream_t s) { return launchHeadsPrec<kF16x3>(a, s); }
hipError_t launchDenseF16x3(const Args& a, hipStream_t s) { return launchDensePrec<kF16x3>(a, s); }

} // namespace tile
} // namespace nsg
