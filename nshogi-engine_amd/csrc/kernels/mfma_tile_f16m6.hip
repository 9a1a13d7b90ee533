// mfma_tile_f16m6.hip -- kF16m6 (f16 main term + e2m3 MX correction terms with per-block E8M0 scales) instantiations of
// the trunk convolution: full tiles only (4 fragments per wave).
#include "mfma_tile.h"

namespace nsg {
namespace tile {

hipError_t launchConvF16m6(const Args& a, int batch, const ConvPlan& p, hipStream_t s) {
    const int gx = (batch + p.nb - 1) / p.nb;
    if (p.nfrag != 4) return hipErrorInvalidValue;
#ifndef NSG_NO_SLAB_SPLIT // (a build without the opt-in slab-split instantiations: make ab ABFLAGS=-DNSG_NO_SLAB_SPLIT)
    // mid batches, two boards per workgroup: the four waves are one 64-channel group whose waves split every chunk
    // pair's slabs four ways, or two groups of two (mfma_tile.h, OwnSeq)
    if (p.sslab == 4 && p.nb == 2 && p.nwaves == 4) return launchOne<kF16m6, kConv, 2, 4, 4, 1, 1, 4>(a, gx, s);
    if (p.sslab == 2 && p.nb == 2 && p.nwaves == 4) return launchOne<kF16m6, kConv, 2, 4, 4, 1, 1, 2>(a, gx, s);
#endif
    if (p.sslab != 1) return hipErrorInvalidValue;
    // mid batches: one board per workgroup, two wave groups on three row fragments each
    // ... or, where all the board's chunk tiles fit in LDS at once, two K halves on all six row fragments
    if (p.ksplit == 4 && p.nb == 1 && p.nwaves == 4) {
        // small batches: four K quarters on one 64-channel group (four workgroups per board); a layer with
        // fewer than four chunk pairs (the stem) runs the two-halves kernel
        // ... and for the very smallest batches each of those workgroups is two: one per half of the rows
        if ((a.kdim / 32) % 8 == 0 && p.msplit == 2) return launchOne<kF16m6, kConv, 1, 4, 4, 2, 4>(a, gx, s);
        if ((a.kdim / 32) % 8 == 0 && p.msplit == 3) return launchOne<kF16m6, kConv, 1, 4, 4, 3, 4>(a, gx, s);
        if ((a.kdim / 32) % 8 == 0 && p.msplit == 6) return launchOne<kF16m6, kConv, 1, 4, 4, 6, 4>(a, gx, s);
        if ((a.kdim / 32) % 8 == 0) return launchOne<kF16m6, kConv, 1, 4, 4, 1, 4>(a, gx, s);
        return launchOne<kF16m6, kConv, 1, 4, 4, 1, 2>(a, gx, s);
    }
    if (p.ksplit == 3 && p.nb == 1 && p.nwaves == 3) {
        // 192 channels = three chunk pairs: one 64-channel group per workgroup (three workgroups per board), its three
        // waves one chunk pair each; the rows over 1, 2, 3 or 6 such workgroups.  A layer that does not have three
        // pairs (the stem: 128 padded input channels) runs two-wave workgroups, a pair per wave.
        if ((a.kdim / 32) % 6 == 0) {
            if (p.msplit == 2) return launchOne<kF16m6, kConv, 1, 4, 3, 2, 3>(a, gx, s);
            if (p.msplit == 3) return launchOne<kF16m6, kConv, 1, 4, 3, 3, 3>(a, gx, s);
            if (p.msplit == 6) return launchOne<kF16m6, kConv, 1, 4, 3, 6, 3>(a, gx, s);
            return launchOne<kF16m6, kConv, 1, 4, 3, 1, 3>(a, gx, s);
        }
        if ((a.kdim / 32) % 4 != 0) return hipErrorInvalidValue;
        if (p.msplit == 2) return launchOne<kF16m6, kConv, 1, 4, 2, 2, 2>(a, gx, s);
        if (p.msplit == 3) return launchOne<kF16m6, kConv, 1, 4, 2, 3, 2>(a, gx, s);
        if (p.msplit == 6) return launchOne<kF16m6, kConv, 1, 4, 2, 6, 2>(a, gx, s);
        return launchOne<kF16m6, kConv, 1, 4, 2, 1, 2>(a, gx, s);
    }
    if (p.ksplit == 2 && p.nb == 1 && p.nwaves == 4) return launchOne<kF16m6, kConv, 1, 4, 4, 1, 2>(a, gx, s);
    if (p.msplit == 2 && p.nb == 1 && p.nwaves == 4) return launchOne<kF16m6, kConv, 1, 4, 4, 2>(a, gx, s);
#define NSG_CASE(NB_, NW_) \
    if (p.nb == NB_ && p.nwaves == NW_) return launchOne<kF16m6, kConv, NB_, 4, NW_>(a, gx, s);
    NSG_CASE(2, 4) NSG_CASE(2, 3) NSG_CASE(2, 2) NSG_CASE(2, 1)
    NSG_CASE(1, 4) NSG_CASE(1, 3) NSG_CASE(1, 2) NSG_CASE(1, 1)
#undef NSG_CASE
    return hipErrorInvalidValue;
}

hipError_t launchTrunkF16m6(const Args* layers, int n, int batch, const ConvPlan& p, hipStream_t s) {
    const int gx = (batch + p.nb - 1) / p.nb;
    if (p.nfrag == 4 && p.nb == 2 && p.nwaves == 4) return launchTrunkOne<kF16m6, 2, 4, 4>(layers, n, gx, s);
    if (p.nfrag == 4 && p.nb == 1 && p.nwaves == 4) return launchTrunkOne<kF16m6, 1, 4, 4>(layers, n, gx, s);
    if (p.nfrag == 4 && p.nb == 2 && p.nwaves == 3) return launchTrunkOne<kF16m6, 2, 4, 3>(layers, n, gx, s);
    if (p.nfrag == 4 && p.nb == 1 && p.nwaves == 3) return launchTrunkOne<kF16m6, 1, 4, 3>(layers, n, gx, s);
    return hipErrorInvalidValue;
}

hipError_t launchCoopTrunkF16m6(const Args* layers, int n, int batch, int cout, const ConvPlan& p, unsigned* flags, unsigned flagBase,
                                int* status, hipStream_t s, int faultBoard) {
    if (p.nb != 1 || p.nfrag != 4 || p.sslab != 1) return hipErrorInvalidValue;
    // the two-way K split of the mid batches: two workgroups per board (128 channels each) on 256 channels
    if (p.nwaves == 4 && p.ksplit == 2 && p.msplit == 1) return launchCoopOne<kF16m6, 4, 4, 1, 2>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
    // the four-way K split of the small batches: four workgroups per board x 1 / 2 / 3 / 6 row groups
    if (p.nwaves == 4 && p.ksplit == 4) {
        if (p.msplit == 1) return launchCoopOne<kF16m6, 4, 4, 1, 4>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
        if (p.msplit == 2) return launchCoopOne<kF16m6, 4, 4, 2, 4>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
        if (p.msplit == 3) return launchCoopOne<kF16m6, 4, 4, 3, 4>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
        if (p.msplit == 6) return launchCoopOne<kF16m6, 4, 4, 6, 4>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
    }
    // 192 channels: the three-way K split, three workgroups per board x row groups
    if (p.nwaves == 3 && p.ksplit == 3) {
        if (p.msplit == 1) return launchCoopOne<kF16m6, 4, 3, 1, 3>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
        if (p.msplit == 2) return launchCoopOne<kF16m6, 4, 3, 2, 3>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
        if (p.msplit == 3) return launchCoopOne<kF16m6, 4, 3, 3, 3>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
        if (p.msplit == 6) return launchCoopOne<kF16m6, 4, 3, 6, 3>(layers, n, batch, cout, flags, flagBase, status, s, faultBoard);
    }
    return hipErrorInvalidValue;
}

} // namespace tile
} // namespace nsg
