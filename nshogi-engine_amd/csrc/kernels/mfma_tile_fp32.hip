// mfma_tile_fp32.hip -- Fp32 instantiations of the MFMA tile kernel
// (one translation unit per precision so the build parallelises).
#include "mfma_tile.h"

namespace nsg {
namespace tile {

hipError_t launchConvFp32(const Args& a, int batch, const ConvPlan& p, hipStream_t s) {
    return launchConvPrec<kFp32>(a, batch, p, s);
}
hipError_t launchTrunkFp32(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s) {
    return launchTrunkPrec<kFp32>(layers, n, batch, p, s);
}
hipError_t launchHeadsFp32(const Args& a, hipStream_t s) { return launchHeadsPrec<kFp32>(a, s); }
hipError_t launchDenseFp32(const Args& a, hipStream_t s) { return launchDensePrec<kFp32>(a, s); }

} // namespace tile
} // namespace nsg
