// mfma_tile_bf16.hip -- Bf16 instantiations of the MFMA tile kernel
// (one translation unit per precision so the build parallelises).
#include "mfma_tile.h"

namespace nsg {
namespace tile {

hipError_t launchConvBf16(const Args& a, int batch, const ConvPlan& p, hipStream_t s) {
    return launchConvPrec<kBf16>(a, batch, p, s);
}
hipError_t launchTrunkBf16(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s) {
    return launchTrunkPrec<kBf16>(layers, n, batch, p, s);
}
hipError_t launchHeadsBf16(const Args& a, hipStream_t s) { return launchHeadsPrec<kBf16>(a, s); }
hipError_t launchDenseBf16(const Args& a, hipStream_t s) { return launchDensePrec<kBf16>(a, s); }

} // namespace tile
} // namespace nsg
