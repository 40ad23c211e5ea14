// Compile-time-geometry instantiations of conv3x3_pws_kernel (conv_pws.h, "compile-time geometry"): the image width W and the tap
// orientation are template parameters, so a pixel fragment's LDS address is a per-lane register + an immediate.  One translation
// unit of their own: they compile beside conv_igemm.hip.
//
// Reference shapes: models/module.py:61-67 (layer2 / layer3 / layer4 of the 128 x 128 student: W = 16 / 8 / 4).
#include <hip/hip_runtime.h>

#include "conv_pws.h"
#include "kernels.h"

namespace {

template <int BM, int BN, int HROWS, int NS, int WC, int NMW = 4, bool PIPE = true>
bool launch_geo(bool flip, const ConvParams& q, const HaloGeom& g, const PwsGrid& sg, dim3 grid, dim3 block, size_t lds,
                hipStream_t stream) {
    const int mode = conv_ep_mode(q);
    if (!flip) {
        switch (mode) {
            case 1: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 1, NMW, PIPE, WC, false>), grid, block, lds, stream, q, g, sg); return true;
            case 3: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 3, NMW, PIPE, WC, false>), grid, block, lds, stream, q, g, sg); return true;
            default: return false;
        }
    }
    switch (mode) {
        case 0: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 0, NMW, PIPE, WC, true>), grid, block, lds, stream, q, g, sg); return true;
        case 2: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 2, NMW, PIPE, WC, true>), grid, block, lds, stream, q, g, sg); return true;
        case 6: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 6, NMW, PIPE, WC, true>), grid, block, lds, stream, q, g, sg); return true;
        case 7: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 7, NMW, PIPE, WC, true>), grid, block, lds, stream, q, g, sg); return true;
        case 8: VPD_LAUNCH((conv3x3_pws_kernel<BM, BN, HROWS, NS, 8, NMW, PIPE, WC, true>), grid, block, lds, stream, q, g, sg); return true;
        default: return false;
    }
}

}  // namespace

bool vpd_launch_pws_geo(int bm, int bn, int hrows, int ns, int nmw, const ConvParams& q, const HaloGeom& g, const PwsGrid& sg, dim3 grid,
                        dim3 block, size_t lds, hipStream_t stream) {
    const TapSet& t = q.taps;
    const bool fwd = t.dy0 == 0 && t.dys == 1 && t.dx0 == 0 && t.dxs == 1;
    const bool flip = t.dy0 == 2 && t.dys == -1 && t.dx0 == 2 && t.dxs == -1;
    if (!fwd && !flip) return false;
    // (the swizzle key the kernel's bases are built from: halo_geom, conv_igemm.hip)
    if (q.Ws >= 8 ? !(g.kmask == 7 && g.kshift == 0 && g.rowmask == 0) : !(g.kmask == 3 && g.kshift == 2 && g.rowmask == 1)) return false;
    if (nmw == 4 && bm == 256 && bn == 64 && hrows == 416 && ns == PWS_NS_C6) {
        if (q.Ws == 16) return launch_geo<256, 64, 416, PWS_NS_C6, 16>(flip, q, g, sg, grid, block, lds, stream);
        if (q.Ws == 8) return launch_geo<256, 64, 416, PWS_NS_C6, 8>(flip, q, g, sg, grid, block, lds, stream);
    } else if (nmw == 4 && bm == 128 && bn == 64 && hrows == 288 && ns == PWS_NS_C3) {
        if (q.Ws == 4) return launch_geo<128, 64, 288, PWS_NS_C3, 4>(flip, q, g, sg, grid, block, lds, stream);
    }
    // (round 6, measured and removed -- profiles/r06_ab_geo8.txt: the same loop on the eight-wave 256 x 128 tile, whose 168-register
    //  budget it does not fit without spilling and whose two MFMA waves per SIMD already cover each other's reads, 676 vs 636 us per step
    //  for its class at 512 crops; with it and the 128 x 128 tile of layer4 the apply forward read 389 k against 393 k crops/s)
    return false;
}
