// mfma_tile_fp16.hip -- Fp16 instantiations of the MFMA tile kernel
// (one translation unit per precision so the build parallelises).
#include "mfma_tile.h"

namespace nsg {
namespace tile {

hipError_t launchConvFp16(const Args& a, int batch, const ConvPlan& p, hipStream_t s) {
    return launchConvPrec<kFp16>(a, batch, p, s);
}
hipError_t launchTrunkFp16(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s) {
    return launchTrunkPrec<kFp16>(layers, n, batch, p, s);
}
hipError_t launchHeadsFp16(const Args& a, hipStream_t s) { return launchHeadsPrec<kFp16>(a, s); }
hipError_t launchDenseFp16(const Args& a, hipStream_t s) { return launchDensePrec<kFp16>(a, s); }

} // namespace tile
} // namespace nsg
