// conv3x3_pws_xf_kernel launches (round 4): a BasicBlock's second 3x3 convolution whose loader waves apply the FIRST
// convolution's BatchNorm + ReLU on the way into LDS (ConvXf, common.h; the loader itself is in conv_pws.h).  The launch
// that used to sit between the two convolutions -- bn_fwd_fused_kernel, ~5 us of fixed cost around 2-5 us of data movement on
// layer2 .. layer4 -- disappears; the activation and its ReLU bit map are still written out (backward reads them), by the
// convolution's blocks.  Train forward only (statistics epilogue); reference path: torchvision BasicBlock.forward
// (conv1 -> bn1 -> relu -> conv2) under model.train(), as driven by train_vpd_model.py:85-118.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "kernels.h"
#include "conv_pws.h"

namespace {

template <int BM, int BN, int HROWS, int NS, int NMW>
hipError_t launch_xf(const ConvParams& p, const HaloGeom& g, const ConvXf& xf, hipStream_t stream) {
    constexpr int WN = BN / 64, WM = NMW / WN;
    constexpr size_t lds0 = (size_t)2 * HROWS * 128 + (size_t)NS * BN * 128 + 1024 + (size_t)3 * WM * BN * 4;
    const size_t lds = lds0 + (size_t)p.Kc * 8;                 // + scale[Kc], shift[Kc]
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    PwsGrid sg;
    sg.MT = (p.M + BM - 1) / BM;
    sg.NT = p.Co / BN;
    int lanes = pws_cu_count() / sg.NT;
    if (lanes < 1) lanes = 1;
    if (lanes > sg.MT) lanes = sg.MT;
    if (lanes >= 8) lanes &= ~7;
    sg.lanes = lanes;
    sg.xcd = lanes % 8 == 0;
    sg.rNT = 1.0f / (float)sg.NT; sg.rlanes = 1.0f / (float)lanes;
    const dim3 grid(lanes * sg.NT), block((NMW + 4) * 64);
#ifdef PWS_STAMPS
    // diagnostic build: s_memtime stamps of launch 30 of each shape, as differences from the block's entry (median / max over blocks)
    static unsigned long long* dstamps = nullptr;
    static int nlaunch = 0;
    if (!dstamps) (void)hipMalloc(&dstamps, 4096 * 16 * 8);
    (void)hipMemsetAsync(dstamps, 0, 4096 * 16 * 8, stream);
    ConvParams q = p;
    q.err = reinterpret_cast<unsigned*>(dstamps);
    VPD_LAUNCH((conv3x3_pws_xf_kernel<BM, BN, HROWS, NS, NMW>), grid, block, lds, stream, q, g, sg, xf);
    if (++nlaunch == 30) {
        static unsigned long long h[4096 * 16];
        const int nb = (int)grid.x;
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, dstamps, (size_t)nb * 16 * 8, hipMemcpyDeviceToHost);
        const char* names[16] = {"entry", "setup done", "first READY", "K loop done", "epilogue issued", "tiles done", "stats flushed",
                                 "sum centre-tap stores", "L entry", "L sum issuing", "L first chunk landed", "L done", "L sum counted wait",
                                 "L sum at barrier", "all epilogues", "table written"};
        fprintf(stderr, "[pws xf stamps <%d,%d> NS %d, %d blocks] cycles from the consumer's entry (median / max over blocks)\n", BM, BN, NS, nb);
        for (int k = 1; k < 16; ++k) {
            std::vector<long long> d;
            for (int b = 0; b < nb; ++b) if (h[b * 16 + k] && h[b * 16]) d.push_back((long long)(h[b * 16 + k] - h[b * 16]));
            if (d.empty()) continue;
            std::sort(d.begin(), d.end());
            fprintf(stderr, "  %-22s %8lld %8lld\n", names[k], d[d.size() / 2], d.back());
        }
    }
#else
    VPD_LAUNCH((conv3x3_pws_xf_kernel<BM, BN, HROWS, NS, NMW>), grid, block, lds, stream, p, g, sg, xf);
#endif
    return hipGetLastError();
}

// 6: 256 x 64 tiles (layer2 / layer3 at 256 crops), 3: 128 x 64 tiles (layer4); 0: the kernel does not take the shape
int xf_class(const ConvParams& p, HaloGeom* g) {
    static const int pws = getenv("VPD_PWS") ? atoi(getenv("VPD_PWS")) : 1;
    if (!pws) return 0;
    if (conv_ep_mode(p) != 1 || p.Kc % 64 != 0 || p.Co % 64 != 0 || p.M <= 0) return 0;
    if ((long)p.N * p.Hs >= VPD_FDIV_MAX) return 0;
    const int kc = vpd_conv_kernel_class(p, g);
    if (kc != 6 && kc != 3) return 0;
    // a tile is made of whole padded images: no halo pixel belongs to two tiles
    if (!(g->multi || g->TR == p.Hs)) return 0;
    const int imgs = g->multi ? g->TR / p.Hs : 1;
    if (imgs > 32 || (long)imgs * p.Hs * p.Ws * p.Kc * 2 >= (1l << 26)) return 0;      // (image index and byte offset share a word)
    const size_t lds0 = kc == 6 ? (size_t)2 * 416 * 128 + 5 * 64 * 128 + 1024 + 3 * 4 * 64 * 4
                                : (size_t)2 * 288 * 128 + 7 * 64 * 128 + 1024 + 3 * 4 * 64 * 4;
    if (lds0 + (size_t)p.Kc * 8 > 160 * 1024) return 0;
    return kc;
}

}  // namespace

// The plan's question: take the XF kernel for this launch?  OFF by default since the BatchNorm launches read their coefficients once
// per thread (same day, +1.6 % on the step): they now cost 7.0-7.7 us on layer3 and 6.2-7.0 on layer4 instead of 9.4 / 7.0, an XF
// launch still costs ~8.5 us more than its plain twin, and what had been +0.3..0.55 % became -0.1 % at 256 crops and -0.45 % at 512
// (profiles/r04_xf_ab.txt, last table).  Every one of the Co / 64 channel tiles of a pixel tile transforms the same input pixels on
// SIMDs whose issue slots an MFMA wave owns; the launch it replaces shrinks with the tensor.  VPD_CONV_XF=1 takes it wherever it
// fits, VPD_CONV_XF_NT bounds the channel tiles (2 = layer2 only: +-0 there).
bool vpd_conv_xf_ok(const ConvParams& p) {
    static const int on = getenv("VPD_CONV_XF") ? atoi(getenv("VPD_CONV_XF")) : 0;
    static const int max_nt = getenv("VPD_CONV_XF_NT") ? atoi(getenv("VPD_CONV_XF_NT")) : 99;
    if (!on || p.Co / 64 > max_nt) return false;
    HaloGeom g;
    return xf_class(p, &g) != 0;
}
// ... and the operator-level entry point's: does the kernel take this shape at all?
bool vpd_conv_xf_fits(const ConvParams& p) {
    HaloGeom g;
    return xf_class(p, &g) != 0;
}

// p.x: the padded activation this launch WRITES (and convolves); xf.z: the dense tensor it reads
hipError_t vpd_launch_conv_xf(const ConvParams& p, const ConvXf& xf0, hipStream_t stream) {
    HaloGeom g;
    const int kc = xf_class(p, &g);
    if (!kc || !xf0.z || !xf0.rows || !xf0.gamma || !xf0.beta || !xf0.mean || !xf0.rstd || !xf0.scale || !xf0.shift)
        return hipErrorInvalidValue;
    ConvXf xf = xf0;
    xf.rHp = 1.0f / (float)(p.Hs + 2);
    static const int ablate = getenv("VPD_XF_ABLATE") ? atoi(getenv("VPD_XF_ABLATE")) : 0;
    xf.ablate = ablate;
    if (kc == 6) return launch_xf<256, 64, 416, 5, 4>(p, g, xf, stream);
    return launch_xf<128, 64, 288, 7, 4>(p, g, xf, stream);
}
