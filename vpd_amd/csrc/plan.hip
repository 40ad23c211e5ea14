// Network plan: owns the static description of the student (ResNet-18/34 BasicBlock or
// ResNet-50/101 / wide Bottleneck encoder + optional motion MLP), the workspace layout, and the
// launch sequences for eval forward, train forward+loss, backward, and the
// weight re-pack.  Exposes the C ABI of include/vpd_hip.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <functional>
#include <vector>

#include "../../include/vpd_hip.h"
#include "common.h"
#include "kernels.h"

static thread_local std::string g_err;
static int fail(const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return -1;
}
#define HCHECK(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return fail(#expr, _e);  \
    } while (0)

extern "C" const char* vpd_last_error(void) { return g_err.c_str(); }
extern "C" const char* vpd_elem_dtype(void) { return VPD_ELEM_NAME; }      // "bf16" (libvpdhip.so) or "fp16" (libvpdhip_f16.so)
extern "C" int vpd_abi_version(void) { return 2; }      // 2: round 5/6 entry points (vpd_op_conv2d_ep, train flag word, 8 timing classes)

namespace {

constexpr float kBnEps = 1e-5f;
constexpr float kBnMomentum = 0.1f;

struct BnInfo {
    int C = 0;
    long long w_off = 0, b_off = 0;      // gamma / beta in the flat param buffer
    long long rm_off = 0, rv_off = 0;    // in the running-stat buffer
    size_t fl_off = 0;                   // float scratch in ws: mean,rstd,scale,shift,coef[3],escale,eshift (9C)
    size_t rows_off = 0;                 // fused passes: this BatchNorm's own accumulator rows [VPD_FUSED_ROWS][2][C] doubles
    size_t sync_off = 0;                 // ... and the grid-barrier words of its fused backward launch
};
struct ConvInfo {
    int Ci = 0, Co = 0, k = 0, stride = 1, pad = 0;
    int Hin = 0, Win = 0, Hout = 0, Wout = 0;
    bool stem = false;
    int Kc = 0, ntaps = 0;
    long long w_off = 0;                 // OIHW offset in flat params/grads
    long long fwd_off = 0, dgr_off = -1; // bf16 element offsets in the weight arena
    long long wg_off = 0;                // fp32 element offset in the wgrad scratch
    long long slab_off = -1;             // fp32 element offset of this conv's split slabs (3x3 s1 convs) or -1
    size_t dz_own_off = 0;               // grouped weight gradients: this conv's own padded dz buffer (kept until the stage's launch)
    long long gslab_off = 0;             // ... and its slab inside the stage's grouped slab (floats)
    BnInfo bn;
    size_t z_off = 0;                    // dense bf16 conv output (train)
};
struct BlockInfo {
    ConvInfo c1, c2, c3, cd;             // c3: Bottleneck's closing 1x1 conv (BasicBlock: unused)
    bool ds = false;
    int stage = 0;
    size_t a1_off = 0, a2_off = 0, out_off = 0;      // padded bf16 activations (a2: Bottleneck only)
    size_t mask_off = 0;                             // train, BasicBlock: [M][C/8] ReLU mask bits of the block output
    size_t mask1_off = 0;                            // train with dgrad_sums: ReLU mask bits of a1 (0: none)
    size_t mask2_off = 0;                            // ... of a2 (Bottleneck students, layer3 / layer4)
};
struct StageInfo {
    int H = 0, W = 0, C = 0;
    size_t dz2_off[2] = {0, 0}, dz1_off[2] = {0, 0}, dzd_off = 0, idn_off = 0;   // dz buffers ping-pong by block parity
    size_t dz3_off = 0;                  // Bottleneck: dz of the closing 1x1 conv
};
struct TensorRow {
    int kind, is_dec;
    long long off, numel;
    int ndim, dims[4];
};
struct LinInfo {
    int in = 0, out = 0;
    long long w_off = 0, b_off = 0;
};

}  // namespace

struct vpd_plan {
    int c_in, H, W, D, motion, max_batch, train;
    int bottleneck = 0, base_width = 64, feat = 512;     // Bottleneck archs: expansion 4, feat = 2048
    std::vector<int> layers;
    ConvInfo stem;
    std::vector<BlockInfo> blocks;
    StageInfo stages[4];
    LinInfo fc, dec[3];
    std::vector<TensorRow> tensors;
    std::vector<BnInfo*> bns;
    long long nparam = 0, nparam_padded = 0, nbn = 0;
    long long arena_elems = 0, wg_elems = 0, slab_elems = 0;
    // gradient buckets (flat-buffer ranges) -- bucket 0 = layer4+fc+decoder ... bucket 3 = stem+layer1
    long long bucket_off[4], bucket_numel[4];
    // ... and the part of the weight-gradient scratch (fp32 elements from wg_off) that holds bucket b's conv gradients, the
    // stem excluded (its row-tap packing is always undone into the flat buffer): vpd_plan_bucket_scratch_range
    long long bucket_wg_off[4], bucket_wg_numel[4];
    // workspace offsets (bytes)
    size_t ws_bytes = 0;
    size_t xin_off = 0, arena_off = 0, wg_off = 0, partial_off = 0, z0_off = 0, p0_off = 0, idx_off = 0;
    size_t g0_off = 0, dz0_off = 0, G_off[3] = {0, 0, 0}, T_off[2] = {0, 0}, slab_off = 0;
    size_t pooled_off = 0, emb_off = 0, h1_off = 0, h2_off = 0, pred_off = 0;
    size_t dpred_off = 0, dh2_off = 0, dh1_off = 0, demb_off = 0, dpooled_off = 0;
    size_t desc_off = 0, bmap_pack_off = 0, bmap_unpack_off[4] = {0, 0, 0, 0};
    int xHp = 0, xWp = 0;
    int H0 = 0, W0 = 0, H1 = 0, W1 = 0;   // stem conv output, pooled output
    // descriptor tables (host copies, uploaded by init_workspace)
    std::vector<PackDesc> descs;
    std::vector<int> bmap_pack;
    std::vector<int> bmap_adam;            // fused AdamW + repack: conv tiles, the stem (one block), plain ranges
    size_t bmap_adam_off = 0;
    int nstem_pack_blocks = 0;             // leading entries of bmap_pack that belong to the stem
    std::vector<int> bmap_unpack[4];
    size_t partial_bytes = 0;
    // captured eval graphs keyed by batch size
    struct Graph { int n; hipGraph_t g; hipGraphExec_t e; };
    std::vector<Graph> graphs;
    void* bound_ws = nullptr;
    bool fused_bn = true;       // one launch per BatchNorm and direction (VPD_FUSED_BN=0: finalize / reduce / apply launches)
    size_t fused_off = 0, fused_bytes = 0;      // rows + barrier words of every BatchNorm: zeroed at the start of each pass
    size_t syncerr_off = 0;                     // sticky counter of grid-barrier time-outs (zeroed by init_workspace only)
    bool wg_group = true;       // per-stage grouped weight gradients (VPD_WG_GROUP=0: one launch per conv)
    size_t gslab_off = 0;       // grouped slab region (bytes offset), sized for the largest launch group
    // lazy gradients (vpd_plan_set_lazy_grads): the next vpd_backward leaves the conv weight gradients in the scratch
    // (only the stem's are unpacked), vpd_plan_adamw_step reads them there; vpd_plan_materialize_grads unpacks on demand
    bool dgrad_sums = true;     // BatchNorm-backward sums in the producing data gradient's epilogue (VPD_DGRAD_SUMS=0: in the BatchNorm launch)
    bool relu_bits = true;      // block-output ReLU masks as bit maps (VPD_RELU_BITS=0: masks from the stored activation, g written back)
    bool lazy_next = false, grads_in_scratch = false;
    int nstem_unpack_blocks = 0;           // leading entries of bmap_unpack[3] that belong to the stem
    bool wg_merge34 = true;     // layer4's grouped weight gradients wait for layer3's and share its launch (VPD_WG_MERGE=0, or the
                                // data-parallel creation flag VPD_TRAIN_EARLY_BUCKET0: per stage)
    bool early_bucket0 = false;
    float loss_scale = 1.f;     // vpd_plan_set_loss_scale: fp16 training (the reference's GradScaler, models/util.py:55-57)
    size_t wg2_tbl_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // task tables of the persistent weight-gradient launches (two per stage)
    void* wg2_cache[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // optional per-kernel-class timing (bench.py roofline): HIP events around every conv launch
    bool timing = false;
    struct TimedLaunch { int cls; double flops; hipEvent_t a, b; };
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> ev_pool;
};

namespace {

struct Bump {
    size_t cur = 0;
    size_t take(size_t bytes) {
        size_t o = cur;
        cur += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

inline size_t padded_elems(int n, int H, int W, int C, int pad) {
    return (size_t)n * (H + 2 * pad) * (W + 2 * pad) * C;
}

void add_tensor(vpd_plan* p, int kind, int is_dec, long long numel, int ndim, int d0, int d1, int d2, int d3,
                long long* off_out) {
    TensorRow r;
    r.kind = kind; r.is_dec = is_dec; r.off = p->nparam; r.numel = numel; r.ndim = ndim;
    r.dims[0] = d0; r.dims[1] = d1; r.dims[2] = d2; r.dims[3] = d3;
    p->tensors.push_back(r);
    *off_out = p->nparam;
    p->nparam += numel;
}

void add_conv(vpd_plan* p, ConvInfo& c, int Ci, int Co, int k, int stride, int pad, int Hin, int Win, bool stem) {
    c.Ci = Ci; c.Co = Co; c.k = k; c.stride = stride; c.pad = pad; c.Hin = Hin; c.Win = Win; c.stem = stem;
    c.Hout = (Hin + 2 * pad - k) / stride + 1;
    c.Wout = (Win + 2 * pad - k) / stride + 1;
    if (stem) { c.Kc = 64; c.ntaps = k; } else { c.Kc = Ci; c.ntaps = k * k; }
    add_tensor(p, 0, 0, (long long)Co * Ci * k * k, 4, Co, Ci, k, k, &c.w_off);
    c.bn.C = Co;
    add_tensor(p, 1, 0, Co, 1, Co, 0, 0, 0, &c.bn.w_off);
    add_tensor(p, 2, 0, Co, 1, Co, 0, 0, 0, &c.bn.b_off);
    c.bn.rm_off = p->nbn; c.bn.rv_off = p->nbn + Co;
    p->nbn += 2 * Co;
    c.fwd_off = p->arena_elems;
    p->arena_elems += (long long)c.ntaps * Co * c.Kc;
    if (!stem) {
        c.dgr_off = p->arena_elems;
        p->arena_elems += (long long)k * k * Ci * Co;
    }
    c.wg_off = p->wg_elems;
    p->wg_elems += (long long)c.ntaps * Co * c.Kc;
    // split slabs of the single (not grouped) halo launches: TWO regions of the maximum size -- region 0 for the stem and the
    // 3x3 convs, region 1 for the 1x1 convs, so that a down-sampling block's two weight gradients can both wait for their
    // stage's slab-reduce launch (round 4); a region is summed before it is written again
    if (stem && p->train && p->slab_elems == 0)
        p->slab_elems += 2 * (long long)(vpd_wgrad_slab_bytes() / 4);
    if (!stem && ((k == 3 && pad == 1) || (k == 1 && pad == 0)) && (stride == 1 || stride == 2) && p->train &&
        vpd_wgrad_halo_shape_ok(c.Hout, c.Wout, stride, Hin, Win)) {
        // halo wgrad conv: ONE shared slab, summed right after each wgrad launch while it is still in the Infinity
        // Cache (per-conv slabs summed once per bucket were measured 4 % slower: 490 MB fall out of the cache)
        c.slab_off = k == 1 ? (long long)(vpd_wgrad_slab_bytes() / 4) : 0;
    }
}

TapSet conv_taps_fwd(const ConvInfo& c) {
    TapSet t;
    if (c.stem) {
        // one tap per kernel row; the 7 column taps x 8 channels are 56 (of 64) contiguous values
        t.nr = c.k; t.nc = 1; t.dy0 = 0; t.dys = 1; t.dx0 = 0; t.dxs = 0; t.w0 = 0; t.wrs = 1; t.wcs = 0;
    } else {
        // input tensors carry a 1-pixel border: padded coord = y*stride + r - pad + 1
        t.nr = c.k; t.nc = c.k; t.dy0 = 1 - c.pad; t.dys = 1; t.dx0 = 1 - c.pad; t.dxs = 1;
        t.w0 = 0; t.wrs = c.k; t.wcs = 1;
    }
    return t;
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" int vpd_plan_create(const char* arch, int c_in, int img_h, int img_w, int emb_dim, int motion,
                               int max_batch, int train, vpd_plan_t** out) {
    if (!arch || !out) return fail("null argument");
    std::vector<int> layers;
    int bottleneck = 0, base_width = 64;          // reference models/module.py:17-32 (ENCODER_ARCH)
    if (!strcmp(arch, "resnet18")) layers = {2, 2, 2, 2};
    else if (!strcmp(arch, "resnet34")) layers = {3, 4, 6, 3};
    else if (!strcmp(arch, "resnet50")) { layers = {3, 4, 6, 3}; bottleneck = 1; }
    else if (!strcmp(arch, "resnet101")) { layers = {3, 4, 23, 3}; bottleneck = 1; }
    else if (!strcmp(arch, "wide_resnet50_2")) { layers = {3, 4, 6, 3}; bottleneck = 1; base_width = 128; }
    else if (!strcmp(arch, "wide_resnet101_2")) { layers = {3, 4, 23, 3}; bottleneck = 1; base_width = 128; }
    else return fail("unsupported arch (resnet18 | resnet34 | resnet50 | resnet101 | wide_resnet50_2 | wide_resnet101_2)");
    if (c_in < 1 || c_in > 8) return fail("c_in must be in 1..8");
    if (img_h < 32 || img_w < 32 || (img_h % 2) || (img_w % 2)) return fail("img dims must be even and >= 32");
    if (emb_dim < 1 || max_batch < 1) return fail("bad emb_dim / max_batch");

    vpd_plan* p = new vpd_plan();
    p->c_in = c_in; p->H = img_h; p->W = img_w; p->D = emb_dim; p->motion = motion ? 1 : 0;
    p->max_batch = max_batch; p->train = train ? 1 : 0; p->layers = layers;
    p->early_bucket0 = (train & VPD_TRAIN_EARLY_BUCKET0) != 0;
    p->bottleneck = bottleneck; p->base_width = base_width; p->feat = bottleneck ? 2048 : 512;

    // ---- topology + flat tables (reference module order) ----
    add_conv(p, p->stem, c_in, 64, 7, 2, 3, img_h, img_w, true);
    p->bns.push_back(&p->stem.bn);
    p->H0 = p->stem.Hout; p->W0 = p->stem.Wout;
    p->H1 = (p->H0 + 2 - 3) / 2 + 1; p->W1 = (p->W0 + 2 - 3) / 2 + 1;
    int inplanes = 64, h = p->H1, w = p->W1;
    const int widths[4] = {64, 128, 256, 512};
    long long stage_first_tensor_off[5];
    int nblocks_total = 0;
    for (int s = 0; s < 4; ++s) nblocks_total += layers[s];
    p->blocks.resize(nblocks_total);
    int bi = 0;
    for (int s = 0; s < 4; ++s) {
        stage_first_tensor_off[s] = p->nparam;
        for (int b = 0; b < layers[s]; ++b, ++bi) {
            BlockInfo& B = p->blocks[bi];
            const int stride = (b == 0 && s > 0) ? 2 : 1;
            B.stage = s;
            if (!bottleneck) {
                add_conv(p, B.c1, inplanes, widths[s], 3, stride, 1, h, w, false);
                add_conv(p, B.c2, widths[s], widths[s], 3, 1, 1, B.c1.Hout, B.c1.Wout, false);
                B.ds = (stride != 1 || inplanes != widths[s]);
                if (B.ds) add_conv(p, B.cd, inplanes, widths[s], 1, stride, 0, h, w, false);
                h = B.c1.Hout; w = B.c1.Wout; inplanes = widths[s];
            } else {
                // torchvision Bottleneck ("v1.5": the 3x3 carries the stride), state_dict order conv1 bn1 conv2 bn2
                // conv3 bn3 downsample.0 downsample.1
                const int width = widths[s] * base_width / 64, outc = widths[s] * 4;
                add_conv(p, B.c1, inplanes, width, 1, 1, 0, h, w, false);
                add_conv(p, B.c2, width, width, 3, stride, 1, h, w, false);
                add_conv(p, B.c3, width, outc, 1, 1, 0, B.c2.Hout, B.c2.Wout, false);
                B.ds = (stride != 1 || inplanes != outc);
                if (B.ds) add_conv(p, B.cd, inplanes, outc, 1, stride, 0, h, w, false);
                h = B.c2.Hout; w = B.c2.Wout; inplanes = outc;
            }
        }
        p->stages[s].H = h; p->stages[s].W = w; p->stages[s].C = inplanes;
    }
    for (auto& B : p->blocks) {     // BN module order: bn1, bn2, (bn3,) downsample.1
        p->bns.push_back(&B.c1.bn);
        p->bns.push_back(&B.c2.bn);
        if (bottleneck) p->bns.push_back(&B.c3.bn);
        if (B.ds) p->bns.push_back(&B.cd.bn);
    }
    if (h < 1 || w < 1) { delete p; return fail("image too small for 5 stride-2 stages"); }
    stage_first_tensor_off[4] = p->nparam;
    p->fc.in = p->feat; p->fc.out = emb_dim;
    add_tensor(p, 3, 0, (long long)emb_dim * p->feat, 2, emb_dim, p->feat, 0, 0, &p->fc.w_off);
    add_tensor(p, 4, 0, emb_dim, 1, emb_dim, 0, 0, 0, &p->fc.b_off);
    if (p->motion) {
        const int dims[4] = {emb_dim, 128, 128, 2 * emb_dim};
        for (int i = 0; i < 3; ++i) {
            p->dec[i].in = dims[i]; p->dec[i].out = dims[i + 1];
            add_tensor(p, 3, 1, (long long)dims[i + 1] * dims[i], 2, dims[i + 1], dims[i], 0, 0, &p->dec[i].w_off);
            add_tensor(p, 4, 1, dims[i + 1], 1, dims[i + 1], 0, 0, 0, &p->dec[i].b_off);
        }
    }
    p->nparam_padded = (p->nparam + 3) & ~3LL;
    // buckets in completion order
    p->bucket_off[0] = stage_first_tensor_off[3]; p->bucket_numel[0] = p->nparam - stage_first_tensor_off[3];
    p->bucket_off[1] = stage_first_tensor_off[2]; p->bucket_numel[1] = stage_first_tensor_off[3] - stage_first_tensor_off[2];
    p->bucket_off[2] = stage_first_tensor_off[1]; p->bucket_numel[2] = stage_first_tensor_off[2] - stage_first_tensor_off[1];
    p->bucket_off[3] = 0; p->bucket_numel[3] = stage_first_tensor_off[1];

    // ---- pack descriptors + block maps ----
    auto push_desc = [&](const ConvInfo& c, int bucket) {
        PackDesc d;
        d.src_off = c.w_off; d.fwd_off = c.fwd_off; d.dgr_off = c.dgr_off; d.wg_off = c.wg_off;
        d.Co = c.Co; d.Ci = c.Ci; d.kh = c.k; d.kw = c.k; d.Kc = c.Kc; d.ntaps = c.ntaps; d.stem = c.stem ? 1 : 0;
        d.numel = 0;
        const int id = (int)p->descs.size();
        p->descs.push_back(d);
        const long long nf = (long long)c.ntaps * c.Co * c.Kc;
        const long long ns = (long long)c.Co * c.Ci * c.k * c.k;
        const long long npk = nf > ns ? nf : ns;
        // pack kernel: 32x32 (co x ci) tiles for ordinary convs, PACK_CHUNK element chunks for the stem
        const long long npack = c.stem ? (npk + 1023) / 1024 : (long long)(c.Co / 32) * (c.Ci / 32);
        for (long long ch = 0; ch < npack; ++ch) { p->bmap_pack.push_back(id); p->bmap_pack.push_back((int)ch); }
        const long long uchunk = 1024;     // PACK_CHUNK of unpack_grads_kernel
        for (long long ch = 0; ch * uchunk < ns; ++ch) {
            p->bmap_unpack[bucket].push_back(id);
            p->bmap_unpack[bucket].push_back((int)ch);
        }
    };
    push_desc(p->stem, 3);
    p->nstem_unpack_blocks = (int)p->bmap_unpack[3].size() / 2;
    for (int b = 0; b < 4; ++b) { p->bucket_wg_off[b] = -1; p->bucket_wg_numel[b] = 0; }
    for (auto& B : p->blocks) {
        const int bucket = 3 - B.stage;
        push_desc(B.c1, bucket);
        push_desc(B.c2, bucket);
        if (bottleneck) push_desc(B.c3, bucket);
        if (B.ds) push_desc(B.cd, bucket);
        // the scratch is laid out in construction order (add_conv), stage after stage: a bucket's convs are ONE range
        auto take = [&](const ConvInfo& cv) {
            const long long n = (long long)cv.ntaps * cv.Co * cv.Kc;
            if (p->bucket_wg_off[bucket] < 0) p->bucket_wg_off[bucket] = cv.wg_off;
            p->bucket_wg_off[bucket] = std::min(p->bucket_wg_off[bucket], (long long)cv.wg_off);
            p->bucket_wg_numel[bucket] += n;
        };
        take(B.c1); take(B.c2);
        if (bottleneck) take(B.c3);
        if (B.ds) take(B.cd);
    }
    {
        // contiguity check: the four ranges tile [stem's end, wg_elems) in bucket order 3, 2, 1, 0
        long long at = (long long)p->stem.ntaps * p->stem.Co * p->stem.Kc;
        for (int b = 3; b >= 0; --b) {
            if (p->bucket_wg_off[b] != at) { delete p; return fail("internal: weight-gradient scratch is not bucket-contiguous"); }
            at += p->bucket_wg_numel[b];
        }
        if (at != p->wg_elems) { delete p; return fail("internal: weight-gradient scratch size mismatch"); }
    }
    {
        // block map of vpd_plan_adamw_step: every conv tile as in the pack map (not the stem), then the ranges
        // of [0, nparam_padded) that are not conv weights, in 2048-float chunks (ADAM_PLAIN_CHUNK of optim.hip)
        const int nconv = (int)p->descs.size();
        // (descs[0] is the stem: a plain range here, packed by a 28-block pack_weights_kernel launch afterwards.  Round 4 tried ONE
        //  block of this launch for it -- update, then the row-tap packing, which gathers across the whole tensor: its ~30 dependent
        //  round trips under the launch's 5 TB/s of traffic made it the launch's pole, 155 vs 131 us, profiles/r04_small_folds.txt)
        for (size_t i = 0; i + 1 < p->bmap_pack.size(); i += 2) {
            if (p->bmap_pack[i] == 0) { p->nstem_pack_blocks++; continue; }
            p->bmap_adam.push_back(p->bmap_pack[i]); p->bmap_adam.push_back(p->bmap_pack[i + 1]);
        }
        std::vector<std::pair<long long, long long>> convs;                   // (offset, numel), ascending
        for (int i = 1; i < nconv; ++i)
            convs.push_back({p->descs[i].src_off, (long long)p->descs[i].Co * p->descs[i].Ci * p->descs[i].kh * p->descs[i].kw});
        std::sort(convs.begin(), convs.end());
        long long pos = 0;
        auto plain = [&](long long a, long long b) {
            if (b <= a) return;
            PackDesc d = {};
            d.src_off = a; d.numel = b - a; d.stem = 2; d.kh = d.kw = 1; d.dgr_off = -1;
            const int id = (int)p->descs.size();
            p->descs.push_back(d);
            for (long long ch = 0; ch * 2048 < b - a; ++ch) { p->bmap_adam.push_back(id); p->bmap_adam.push_back((int)ch); }
        };
        for (auto& cv : convs) { plain(pos, cv.first); pos = cv.first + cv.second; }
        plain(pos, p->nparam_padded);
    }

    // ---- workspace layout ----
    Bump bp;
    const int NB = max_batch;
    p->xHp = img_h + 6; p->xWp = img_w + 8;
    p->xin_off = bp.take(((size_t)NB * p->xHp * p->xWp * 8 + 256) * 2);
    p->arena_off = bp.take((size_t)p->arena_elems * 2);
    p->wg_off = bp.take((size_t)p->wg_elems * 4);
    for (BnInfo* b : p->bns) b->fl_off = bp.take((size_t)9 * b->C * 4);
    p->fused_bn = !(getenv("VPD_FUSED_BN") && !atoi(getenv("VPD_FUSED_BN")));
    {
        const size_t start = bp.cur;
        for (BnInfo* b : p->bns) {
            b->rows_off = bp.take((size_t)VPD_FUSED_ROWS * 2 * b->C * sizeof(double));
            b->sync_off = bp.take(VPD_GRID_SYNC_BYTES);
        }
        p->fused_off = start; p->fused_bytes = bp.cur - start;      // (Bump rounds every piece to 256 B: 16-byte multiples)
        p->syncerr_off = bp.take(256);
    }
    p->desc_off = bp.take(p->descs.size() * sizeof(PackDesc));
    p->bmap_pack_off = bp.take(p->bmap_pack.size() * sizeof(int));
    p->bmap_adam_off = bp.take(p->bmap_adam.size() * sizeof(int));
    for (int i = 0; i < 4; ++i) p->bmap_unpack_off[i] = bp.take(p->bmap_unpack[i].size() * sizeof(int) + 16);
    // statistics partials: max over layers of T*2*C floats
    {
        size_t mx = 0;
        auto upd = [&](const ConvInfo& c) {
            const long long M = (long long)NB * c.Hout * c.Wout;
            const int bm = vpd_conv_bm((int)M, c.Co);
            size_t t1 = (size_t)((M + bm - 1) / bm) * 2 * c.Co * 4;
            int ppb;
            size_t t2 = (size_t)vpd_bn_bwd_blocks((int)M, c.Co, &ppb) * 2 * c.Co * 4;
            // small batches use smaller BM choices: be generous
            size_t t3 = (size_t)((M + 63) / 64) * 2 * c.Co * 4;
            mx = t1 > mx ? t1 : mx; mx = t2 > mx ? t2 : mx; mx = t3 > mx ? t3 : mx;
        };
        upd(p->stem);
        for (auto& B : p->blocks) { upd(B.c1); upd(B.c2); if (B.ds) upd(B.cd); }
        (void)mx;   // producers accumulate atomically into VPD_STAT_ROWS rows of [2][C]
        p->partial_bytes = (size_t)VPD_STAT_ROWS * 2 * p->feat * sizeof(double);
        p->partial_off = bp.take(p->partial_bytes);
    }
    p->z0_off = bp.take((size_t)NB * p->H0 * p->W0 * 64 * 2);
    p->p0_off = bp.take(padded_elems(NB, p->H1, p->W1, 64, 1) * 2);
    for (auto& B : p->blocks) {
        B.a1_off = bp.take(padded_elems(NB, B.c1.Hout, B.c1.Wout, B.c1.Co, 1) * 2);
        if (bottleneck) {
            B.a2_off = bp.take(padded_elems(NB, B.c2.Hout, B.c2.Wout, B.c2.Co, 1) * 2);
            B.out_off = bp.take(padded_elems(NB, B.c3.Hout, B.c3.Wout, B.c3.Co, 1) * 2);
        } else {
            B.out_off = bp.take(padded_elems(NB, B.c1.Hout, B.c1.Wout, B.c1.Co, 1) * 2);
        }
    }
    for (int s = 0; s < 4; ++s) {
        StageInfo& S = p->stages[s];
        S.idn_off = bp.take(padded_elems(NB, S.H, S.W, S.C, 1) * 2);
    }
    p->pooled_off = bp.take((size_t)NB * p->feat * 4);
    p->emb_off = bp.take((size_t)NB * emb_dim * 4);
    p->h1_off = bp.take((size_t)NB * 128 * 4);
    p->h2_off = bp.take((size_t)NB * 128 * 4);
    p->pred_off = bp.take((size_t)NB * 2 * emb_dim * 4);
    if (p->train) {
        p->idx_off = bp.take((size_t)NB * p->H1 * p->W1 * 64);
        size_t maxact = (size_t)NB * p->H1 * p->W1 * 64;
        for (auto& B : p->blocks) {
            B.c1.z_off = bp.take((size_t)NB * B.c1.Hout * B.c1.Wout * B.c1.Co * 2);
            B.c2.z_off = bp.take((size_t)NB * B.c2.Hout * B.c2.Wout * B.c2.Co * 2);
            if (bottleneck) B.c3.z_off = bp.take((size_t)NB * B.c3.Hout * B.c3.Wout * B.c3.Co * 2);
            if (B.ds) B.cd.z_off = bp.take((size_t)NB * B.cd.Hout * B.cd.Wout * B.cd.Co * 2);
            size_t e = (size_t)NB * B.c1.Hout * B.c1.Wout * B.c1.Co;
            maxact = e > maxact ? e : maxact;
            if (bottleneck) {      // gradients w.r.t. the block input / output and both inner activations
                e = (size_t)NB * B.c1.Hin * B.c1.Win * B.c1.Ci; maxact = e > maxact ? e : maxact;
                e = (size_t)NB * B.c2.Hout * B.c2.Wout * B.c2.Co; maxact = e > maxact ? e : maxact;
                e = (size_t)NB * B.c3.Hout * B.c3.Wout * B.c3.Co; maxact = e > maxact ? e : maxact;
            }
        }
        for (int s = 0; s < 4; ++s) {
            StageInfo& S = p->stages[s];
            for (int k = 0; k < 2; ++k) {
                S.dz2_off[k] = bp.take(padded_elems(NB, S.H, S.W, S.C, 1) * 2);
                S.dz1_off[k] = bp.take(padded_elems(NB, S.H, S.W, S.C, 1) * 2);
            }
            S.dzd_off = bp.take(padded_elems(NB, S.H, S.W, S.C, 1) * 2);
        }
        if (bottleneck) {      // per-stage dz buffers sized for the largest conv output of the stage's blocks
            for (int s = 0; s < 4; ++s) {
                size_t m1 = 0, m2 = 0, m3 = 0;
                for (auto& B : p->blocks) {
                    if (B.stage != s) continue;
                    size_t e1 = padded_elems(NB, B.c1.Hout, B.c1.Wout, B.c1.Co, 1), e2 = padded_elems(NB, B.c2.Hout, B.c2.Wout, B.c2.Co, 1),
                           e3 = padded_elems(NB, B.c3.Hout, B.c3.Wout, B.c3.Co, 1);
                    m1 = e1 > m1 ? e1 : m1; m2 = e2 > m2 ? e2 : m2; m3 = e3 > m3 ? e3 : m3;
                }
                StageInfo& S = p->stages[s];
                S.dz1_off[0] = S.dz1_off[1] = bp.take(m1 * 2);
                S.dz2_off[0] = S.dz2_off[1] = bp.take(m2 * 2);
                S.dz3_off = bp.take(m3 * 2);
            }
            for (int i = 0; i < 2; ++i) p->T_off[i] = bp.take(maxact * 2);
        }
        // grouped weight gradients: every eligible 3x3 stride-1 conv keeps its own dz until the
        // stage's grouped launch; the stage's slab holds every problem's splits at once
        p->wg_group = !(getenv("VPD_WG_GROUP") && !atoi(getenv("VPD_WG_GROUP")));
        // data parallel (VPD_TRAIN_EARLY_BUCKET0): layer4's weight gradients get a launch of their own at the end of layer4's
        // backward, so that bucket 0 -- fc + layer4, 61 % of the gradient bytes -- is handed to the reducer there instead of
        // together with bucket 1 behind layer3 (single GPU: the merged launch fills the chip, +0.5 % on the step)
        p->wg_merge34 = !(getenv("VPD_WG_MERGE") && !atoi(getenv("VPD_WG_MERGE"))) && !p->early_bucket0;
        if (p->wg_group) {
            // slabs of one LAUNCH live side by side: with wg_merge34 the stages 2 and 3 (layer3, layer4) share a launch
            size_t stage_slab[4] = {0, 0, 0, 0};
            auto wq = [&](const ConvInfo& cv) {
                WgradParams q;
                memset(&q, 0, sizeof q);
                q.dzHp = cv.Hout + 2; q.dzWp = cv.Wout + 2; q.dzC = cv.Co; q.dzpad = 1;
                q.xHp = cv.Hin + 2; q.xWp = cv.Win + 2; q.xC = cv.Ci;
                q.N = NB; q.Hs = cv.Hout; q.Ws = cv.Wout; q.istr = cv.stride; q.Kc = cv.Kc; q.Co = cv.Co;
                q.M = NB * cv.Hout * cv.Wout; q.taps = conv_taps_fwd(cv);
                return q;
            };
            auto wq_swapped = [&](const ConvInfo& cv) {      // dz := the input activation, x := dz (WgradParams::transposed)
                WgradParams q = wq(cv);
                if (cv.k != 1 || cv.stride != 1 || cv.Co % 128 == 0) { q.Co = 0; return q; }      // (Co = 0: never eligible)
                q.dzC = cv.Ci; q.xC = cv.Co; q.Co = cv.Ci; q.Kc = cv.Co; q.transposed = 1;
                return q;
            };
            for (auto& B : p->blocks) {
                std::vector<ConvInfo*> cvs = {&B.c1, &B.c2};
                if (bottleneck) cvs.push_back(&B.c3);
                if (B.ds) cvs.push_back(&B.cd);
                for (ConvInfo* cv : cvs) {
                    if (cv->slab_off < 0) continue;
                    if (cv->k == 1) {
                        // 1x1 convolutions: tasks of the stage's persistent launch when it takes them (Co % 128 == 0 ...).
                        // Bottleneck students: ResNet-50 step 9.32 -> 8.74 ms (36 launches of the atomics kernel at 47 us each
                        // become tasks; no atomics left).  BasicBlock students keep their three down-sampling convs on
                        // launches of their own (same-box: 3.945 vs 3.951 ms grouped).
                        // (a conv with < 128 output but >= 128 input channels -- layer1's 256 -> 64 -- joins with its operands swapped)
                        if (!bottleneck || !(vpd_wgrad128_eligible(wq(*cv)) || vpd_wgrad128_eligible(wq_swapped(*cv)))) continue;
                    } else if (cv->k != 3) {
                        continue;
                    } else if (cv->stride != 1) {
                        // stride-2 3x3: launches of their own (inside the stage's launch the three stride-2 problems cost 86 us
                        // per step against 78 us: their 55-66 KB per chunk leave room for a two-stage ring only)
                        continue;
                    }
                    cv->dz_own_off = bp.take(padded_elems(NB, cv->Hout, cv->Wout, cv->Co, 1) * 2);
                    const int grp = (p->wg_merge34 && B.stage == 3) ? 2 : B.stage;
                    cv->gslab_off = (long long)stage_slab[grp];
                    stage_slab[grp] += vpd_wgrad_group_slab_floats(NB * cv->Hout * cv->Wout, cv->Co, cv->Kc, cv->k == 1 ? 1 : 9);
                }
            }
            size_t mx = 16;
            for (int s2 = 0; s2 < 4; ++s2) mx = stage_slab[s2] > mx ? stage_slab[s2] : mx;
            p->gslab_off = bp.take(mx * 4);
            for (int s2 = 0; s2 < 8; ++s2) p->wg2_tbl_off[s2] = bp.take(vpd_wgrad128_table_bytes());
        }
        p->relu_bits = !(getenv("VPD_RELU_BITS") && !atoi(getenv("VPD_RELU_BITS")));
        if (p->relu_bits)
            for (auto& B : p->blocks) {
                const ConvInfo& last = bottleneck ? B.c3 : B.c2;      // the conv whose BatchNorm feeds the block-output ReLU
                B.mask_off = bp.take((size_t)NB * last.Hout * last.Wout * last.Co / 8 + 16);
            }
        // BasicBlock students: every block.  Bottleneck students: layer3 / layer4 only, the BatchNorms of conv1 / conv2 -- there the
        // backward launches sit on the grid barrier's latency chain (14-16 us for 2-8 MB) like a BasicBlock student's; in layer1 /
        // layer2 the tensors are 4-16x larger, the launches run at the memory system's rate, and the second read of z in the data
        // gradients' epilogues costs what the shorter BatchNorm launch saves (ResNet-50 8.73 -> 8.76 ms with all stages, rounds 2 / 4)
        p->dgrad_sums = p->relu_bits && p->fused_bn && !(getenv("VPD_DGRAD_SUMS") && !atoi(getenv("VPD_DGRAD_SUMS")));
        if (p->dgrad_sums)
            for (auto& B : p->blocks) {
                if (bottleneck && B.stage < 2) continue;
                B.mask1_off = bp.take((size_t)NB * B.c1.Hout * B.c1.Wout * B.c1.Co / 8 + 16);
                if (bottleneck) B.mask2_off = bp.take((size_t)NB * B.c2.Hout * B.c2.Wout * B.c2.Co / 8 + 16);
            }
        for (int i = 0; i < 3; ++i) p->G_off[i] = bp.take(maxact * 2);
        p->slab_off = bp.take((size_t)(p->slab_elems > 0 ? p->slab_elems : 1) * 4);
        p->g0_off = bp.take((size_t)NB * p->H0 * p->W0 * 64 * 2);
        p->dz0_off = bp.take((size_t)NB * p->H0 * p->W0 * 64 * 2);
        p->dpred_off = bp.take((size_t)NB * 2 * emb_dim * 4);
        p->dh2_off = bp.take((size_t)NB * 128 * 4);
        p->dh1_off = bp.take((size_t)NB * 128 * 4);
        p->demb_off = bp.take((size_t)NB * emb_dim * 4);
        p->dpooled_off = bp.take((size_t)NB * p->feat * 4);
    }
    p->ws_bytes = bp.cur;
    *out = p;
    return 0;
}

extern "C" void vpd_plan_destroy(vpd_plan_t* p) {
    if (!p) return;
    for (auto& g : p->graphs) {
        (void)hipGraphExecDestroy(g.e);
        (void)hipGraphDestroy(g.g);
    }
    for (auto& t : p->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : p->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 8; ++i)
        if (p->wg2_cache[i]) vpd_wgrad128_cache_free(p->wg2_cache[i]);
    delete p;
}

extern "C" int vpd_plan_num_tensors(const vpd_plan_t* p) { return (int)p->tensors.size(); }
extern "C" int vpd_plan_tensor_info(const vpd_plan_t* p, int i, int* kind, int* is_decoder, long long* offset,
                                    long long* numel, int* ndim, int dims[4]) {
    if (i < 0 || i >= (int)p->tensors.size()) return fail("tensor index out of range");
    const TensorRow& r = p->tensors[i];
    *kind = r.kind; *is_decoder = r.is_dec; *offset = r.off; *numel = r.numel; *ndim = r.ndim;
    for (int k = 0; k < 4; ++k) dims[k] = r.dims[k];
    return 0;
}
extern "C" long long vpd_plan_param_numel(const vpd_plan_t* p) { return p->nparam_padded; }
extern "C" int vpd_plan_num_bn(const vpd_plan_t* p) { return (int)p->bns.size(); }
extern "C" int vpd_plan_bn_info(const vpd_plan_t* p, int i, int* channels, long long* rm_off, long long* rv_off) {
    if (i < 0 || i >= (int)p->bns.size()) return fail("bn index out of range");
    *channels = p->bns[i]->C; *rm_off = p->bns[i]->rm_off; *rv_off = p->bns[i]->rv_off;
    return 0;
}
extern "C" long long vpd_plan_bn_numel(const vpd_plan_t* p) { return p->nbn; }
extern "C" int vpd_plan_num_buckets(const vpd_plan_t*) { return 4; }
extern "C" int vpd_plan_bucket_range(const vpd_plan_t* p, int b, long long* offset, long long* numel) {
    if (b < 0 || b >= 4) return fail("bucket index out of range");
    *offset = p->bucket_off[b]; *numel = p->bucket_numel[b];
    return 0;
}
extern "C" int vpd_plan_bucket_scratch_range(const vpd_plan_t* p, int b, long long* ws_byte_offset, long long* numel) {
    if (b < 0 || b >= 4) return fail("bucket index out of range");
    if (!p->train) return fail("plan was created with train=0");
    *ws_byte_offset = (long long)p->wg_off + p->bucket_wg_off[b] * 4;
    *numel = p->bucket_wg_numel[b];
    return 0;
}
extern "C" size_t vpd_plan_workspace_bytes(const vpd_plan_t* p) { return p->ws_bytes; }

extern "C" int vpd_plan_init_workspace(vpd_plan_t* p, void* ws, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    HCHECK(hipMemsetAsync(ws, 0, p->ws_bytes, s));
    HCHECK(hipMemcpyAsync(w + p->desc_off, p->descs.data(), p->descs.size() * sizeof(PackDesc), hipMemcpyHostToDevice, s));
    HCHECK(hipMemcpyAsync(w + p->bmap_pack_off, p->bmap_pack.data(), p->bmap_pack.size() * sizeof(int), hipMemcpyHostToDevice, s));
    HCHECK(hipMemcpyAsync(w + p->bmap_adam_off, p->bmap_adam.data(), p->bmap_adam.size() * sizeof(int), hipMemcpyHostToDevice, s));
    for (int i = 0; i < 4; ++i)
        if (!p->bmap_unpack[i].empty())
            HCHECK(hipMemcpyAsync(w + p->bmap_unpack_off[i], p->bmap_unpack[i].data(),
                                  p->bmap_unpack[i].size() * sizeof(int), hipMemcpyHostToDevice, s));
    HCHECK(hipStreamSynchronize(s));   // host vectors may be re-read only now; one-time setup
    p->bound_ws = ws;
    return 0;
}

// ---------------------------------------------------------------------------
namespace {

struct Ctx {
    vpd_plan* p;
    char* ws;
    hipStream_t s;
    const float* params;
    int n;
    bf16_t* b16(size_t off) const { return reinterpret_cast<bf16_t*>(ws + off); }
    float* f32(size_t off) const { return reinterpret_cast<float*>(ws + off); }
    double* stat_rows() const { return reinterpret_cast<double*>(ws + p->partial_off); }
    double* bn_rows(const BnInfo& b) const { return reinterpret_cast<double*>(ws + b.rows_off); }
    bool fused(const ConvInfo& cv) const { return p->fused_bn && !cv.stem; }
    float* bn_mean(const BnInfo& b) const { return f32(b.fl_off); }
    float* bn_rstd(const BnInfo& b) const { return f32(b.fl_off) + b.C; }
    float* bn_scale(const BnInfo& b) const { return f32(b.fl_off) + 2 * b.C; }
    float* bn_shift(const BnInfo& b) const { return f32(b.fl_off) + 3 * b.C; }
    float* bn_coef(const BnInfo& b) const { return f32(b.fl_off) + 4 * b.C; }
    float* bn_escale(const BnInfo& b) const { return f32(b.fl_off) + 7 * b.C; }
    float* bn_eshift(const BnInfo& b) const { return f32(b.fl_off) + 8 * b.C; }
};

// timing classes: 0..4 = vpd_conv_kernel_class, 5 = conv_wgrad_halo_kernel (without its slab reduce), 6 = conv_wgrad_kernel
struct TimeScope {
    vpd_plan* p; hipStream_t s; int idx = -1;
    TimeScope(vpd_plan* p_, hipStream_t s_, int cls, double flops) : p(p_), s(s_) {
        if (!p->timing) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!p->ev_pool.empty()) { e = p->ev_pool.back(); p->ev_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        vpd_plan::TimedLaunch t{cls, flops, get(), get()};
        p->timed.push_back(t);
        idx = (int)p->timed.size() - 1;
        vpd_launch_events() = {t.a, t.b};       // the scope's first matrix-kernel launch carries them (common.h)
    }
    ~TimeScope() {
        if (idx < 0) return;
        if (vpd_launch_events().start) {        // nothing was launched through VPD_LAUNCH: bracket the scope instead
            vpd_launch_events().start = nullptr;
            (void)hipEventRecord(p->timed[idx].a, s);
            (void)hipEventRecord(p->timed[idx].b, s);
        }
    }
};
inline double conv_flops(const ConvInfo& cv, int n) {      // algorithmic: real taps and channels
    return 2.0 * n * cv.Hout * cv.Wout * cv.Co * (double)cv.Ci * cv.k * cv.k;
}

// forward convolution launch; input padded activation `x` (border 1; stem: xin), output `y`
// second convolution of the same launch (ConvParams::alt_*): a BasicBlock's 1x1 down-sampling branch beside its first 3x3
struct AltConv { const ConvInfo* cv; bf16_t* y; const float* ep_scale; const float* ep_shift; int ep_relu; };
// can `cd` ride in `c1`'s launch?  Same input, same output geometry and channel count; train mode needs per-BatchNorm
// statistics rows (the shared rows serve one conv at a time).  VPD_DS_MERGE=0 keeps the two launches.
bool conv_pair_ok(const Ctx& c, const ConvInfo& c1, const ConvInfo& cd, bool train) {
    static const bool off = getenv("VPD_DS_MERGE") && !atoi(getenv("VPD_DS_MERGE"));
    if (off || c.p->bottleneck || c1.k != 3 || cd.k != 1 || c1.stride != 2 || cd.stride != 2) return false;
    if (c1.Hin != cd.Hin || c1.Win != cd.Win || c1.Hout != cd.Hout || c1.Wout != cd.Wout || c1.Ci != cd.Ci || c1.Co != cd.Co)
        return false;
    return !train || (c.fused(c1) && c.fused(cd));
}

// pool_y / pooled: the eval stem with scale / shift / ReLU / max-pool in the conv's epilogue (ConvParams::pool_y) when the stem
// kernel takes the shape; *pooled tells the caller whether it did (false: plain conv into y, the pooling launch follows)
hipError_t run_conv_fwd(const Ctx& c, const ConvInfo& cv, const bf16_t* x, bf16_t* y, int ypad, bool stats,
                        const float* ep_scale, const float* ep_shift, const bf16_t* res, int ep_relu,
                        const AltConv* alt = nullptr, bf16_t* pool_y = nullptr, bool* pooled = nullptr) {
    ConvParams q;
    memset(&q, 0, sizeof q);
    q.x = x;
    if (cv.stem) { q.xHp = c.p->xHp; q.xWp = c.p->xWp; q.xC = 8; }
    else { q.xHp = cv.Hin + 2; q.xWp = cv.Win + 2; q.xC = cv.Ci; }
    q.w = c.b16(c.p->arena_off) + cv.fwd_off;
    q.y = y; q.yHp = cv.Hout + 2 * ypad; q.yWp = cv.Wout + 2 * ypad; q.yC = cv.Co; q.ypad = ypad;
    q.stats = stats ? (c.fused(cv) ? c.bn_rows(cv.bn) : c.stat_rows()) : nullptr;
    q.stat_rows = c.fused(cv) ? VPD_FUSED_ROWS : 0;
    q.ep_scale = ep_scale; q.ep_shift = ep_shift; q.res = res; q.ep_relu = ep_relu;
    q.rHp = cv.Hout + 2; q.rWp = cv.Wout + 2; q.rC = cv.Co; q.rpad = 1;
    q.N = c.n; q.Hs = cv.Hout; q.Ws = cv.Wout; q.osub = 1; q.oph = 0; q.opw = 0; q.istr = cv.stride;
    q.Kc = cv.Kc; q.Co = cv.Co; q.M = c.n * cv.Hout * cv.Wout; q.accumulate = 0;
    q.taps = conv_taps_fwd(cv);
    q.err = reinterpret_cast<unsigned*>(c.ws + c.p->syncerr_off);
    double flops = conv_flops(cv, c.n);
    if (alt) {
        const ConvInfo& av = *alt->cv;
        q.alt_w = c.b16(c.p->arena_off) + av.fwd_off; q.alt_y = alt->y; q.alt_taps = conv_taps_fwd(av);
        q.alt_stats = stats ? c.bn_rows(av.bn) : nullptr;
        q.alt_ep_scale = alt->ep_scale; q.alt_ep_shift = alt->ep_shift; q.alt_ep_relu = alt->ep_relu;
        flops += conv_flops(av, c.n);
    }
    if (pool_y) {
        q.pool_y = pool_y;
        const bool ok = vpd_conv_kernel_class(q) == 5;
        if (!ok) { q.pool_y = nullptr; q.ep_scale = nullptr; q.ep_shift = nullptr; q.ep_relu = 0; }
        if (pooled) *pooled = ok;
    }
    const int kc = vpd_conv_kernel_class(q);
    // slot 7: stem kernel (5, 6 are the wgrads); ws<256,64> shares slot 2 -- except layer1's 64 -> 64 convs, which stay in slot 0
    const int tcls = kc == 5 ? 7 : (kc == 6 ? (cv.Co == 64 && cv.Ci == 64 ? 0 : 2) : kc);
    TimeScope ts(c.p, c.s, tcls, flops);
    return vpd_launch_conv(q, c.s);
}

hipError_t run_bn_finalize(const Ctx& c, const ConvInfo& cv, float* bn_running) {
    const int M = c.n * cv.Hout * cv.Wout;
    const int bm = vpd_conv_bm(M, cv.Co);
    (void)bm;
    const int T = VPD_STAT_ROWS;     // unused accumulator rows are zero; the producer's tile size is its own business
    return vpd_launch_bn_finalize(c.stat_rows(), T, cv.Co, (float)M, c.params + cv.bn.w_off,
                                  c.params + cv.bn.b_off, bn_running ? bn_running + cv.bn.rm_off : nullptr,
                                  bn_running ? bn_running + cv.bn.rv_off : nullptr, kBnMomentum, kBnEps,
                                  c.bn_mean(cv.bn), c.bn_rstd(cv.bn), c.bn_scale(cv.bn), c.bn_shift(cv.bn), c.s);
}

// data-gradient of a conv: dz (padded, border 1) -> dx (dense [n][Hin][Win][Ci])
int device_cu_count() { return vpd_cu_budget(); }

// stride-1 data-gradient launch descriptor
ConvParams conv_dgrad_s1_params(const Ctx& c, const ConvInfo& cv, const bf16_t* dz, bf16_t* dx, int accumulate) {
    ConvParams q;
    memset(&q, 0, sizeof q);
    q.x = dz; q.xHp = cv.Hout + 2; q.xWp = cv.Wout + 2; q.xC = cv.Co;
    q.w = c.b16(c.p->arena_off) + cv.dgr_off;
    q.y = dx; q.yHp = cv.Hin; q.yWp = cv.Win; q.yC = cv.Ci; q.ypad = 0;
    q.N = c.n; q.Kc = cv.Co; q.Co = cv.Ci; q.accumulate = accumulate; q.istr = 1;
    // dx[y][x] = sum_{r,t} dz[y + pad - r][x + pad - t] W[r][t]; padded coord adds 1
    q.Hs = cv.Hin; q.Ws = cv.Win; q.osub = 1; q.oph = 0; q.opw = 0;
    q.M = c.n * q.Hs * q.Ws;
    q.taps.nr = cv.k; q.taps.nc = cv.k;
    q.taps.dy0 = cv.pad + 1; q.taps.dys = -1; q.taps.dx0 = cv.pad + 1; q.taps.dxs = -1;
    q.taps.w0 = 0; q.taps.wrs = cv.k; q.taps.wcs = 1;
    q.err = reinterpret_cast<unsigned*>(c.ws + c.p->syncerr_off);
    return q;
}

// The sums of a BatchNorm backward (sum g, sum g * z with g = d * mask) taken in the epilogue of the data gradient that
// produces d (ConvParams::bst_z); the BatchNorm launch is then finalize + apply only (run_bn_bwd_apply).
// z2 / rows2: a second BatchNorm fed with the same g (the 1x1 branch of a down-sampling block), or null
struct BnSums { const bf16_t* z; const unsigned char* mask; double* rows; const bf16_t* z2; double* rows2; };
bool dgrad_takes_sums(const Ctx& c, const ConvInfo& cv, int accumulate, bool pair = false) {
    if (!c.p->dgrad_sums) return false;
    // a stride-2 conv's merged parity classes (plain store; even input dims: the classes tile the input exactly)
    static const bool s2on = !(getenv("VPD_DGRAD_SUMS_S2") && !atoi(getenv("VPD_DGRAD_SUMS_S2")));      // (as vpd_conv_takes_bn_sums)
    if (cv.stride != 1) return s2on && cv.stride == 2 && cv.k == 3 && !accumulate && !pair && cv.Hin % 2 == 0 && cv.Win % 2 == 0;
    ConvParams q = conv_dgrad_s1_params(c, cv, c.b16(0), c.b16(0), accumulate);
    q.bst_z = c.b16(0);
    if (pair) { q.bst_z2 = c.b16(0); q.stats2 = c.stat_rows(); }
    return vpd_conv_takes_bn_sums(q);
}

// ds / dzd: the block's 1x1 stride-2 down-sampling conv and its dz -- its data gradient lands on the even-even input pixels,
// which are class 0 of the 3x3's: extra K-steps of those blocks instead of a read-modify-write launch of its own
hipError_t run_conv_dgrad(const Ctx& c, const ConvInfo& cv, const bf16_t* dz, bf16_t* dx, int accumulate,
                          const ConvInfo* ds = nullptr, const bf16_t* dzd = nullptr,
                          const unsigned char* acc_mask = nullptr, const BnSums* sums = nullptr) {
    ConvParams q;
    memset(&q, 0, sizeof q);
    q.x = dz; q.xHp = cv.Hout + 2; q.xWp = cv.Wout + 2; q.xC = cv.Co;
    q.w = c.b16(c.p->arena_off) + cv.dgr_off;
    q.y = dx; q.yHp = cv.Hin; q.yWp = cv.Win; q.yC = cv.Ci; q.ypad = 0;
    q.N = c.n; q.Kc = cv.Co; q.Co = cv.Ci; q.accumulate = accumulate; q.istr = 1;
    if (cv.stride == 1) {
        q = conv_dgrad_s1_params(c, cv, dz, dx, accumulate);
        q.acc_mask = accumulate ? acc_mask : nullptr;
        if (sums) {      // (the caller has checked dgrad_takes_sums)
            q.bst_z = sums->z; q.bst_mask = sums->mask;
            q.stats = sums->rows; q.stat_rows = VPD_FUSED_ROWS;
            q.bst_z2 = sums->z2; q.stats2 = sums->rows2;
        }
        const int kcd = vpd_conv_kernel_class(q);
        TimeScope ts(c.p, c.s, kcd == 6 ? (cv.Co == 64 && cv.Ci == 64 ? 0 : 2) : kcd, conv_flops(cv, c.n));
        return vpd_launch_conv(q, c.s);
    }
    // stride 2: the four input-pixel parity classes are ONE launch (grid.z = class).  Only taps r with
    // (ph + pad - r) even contribute: r = rf, rf+2, ... reading dz row  y + (ph + pad - r)/2  (+1 for the border).
    TimeScope ts(c.p, c.s, 4, conv_flops(cv, c.n) + (ds ? conv_flops(*ds, c.n) : 0.0));
    q.osub = 2;
    int ncls = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            ConvClass k;
            k.geo.Hs = (cv.Hin - ph + 1) / 2; k.geo.Ws = (cv.Win - pw + 1) / 2;
            if (k.geo.Hs <= 0 || k.geo.Ws <= 0) continue;
            k.geo.oph = ph; k.geo.opw = pw;
            k.geo.M = c.n * k.geo.Hs * k.geo.Ws;
            const int rf = (ph + cv.pad) % 2, tf = (pw + cv.pad) % 2;
            k.taps.nr = rf < cv.k ? (cv.k - rf + 1) / 2 : 0;
            k.taps.nc = tf < cv.k ? (cv.k - tf + 1) / 2 : 0;
            if (k.taps.nr == 0 || k.taps.nc == 0) continue;   // caller zero-fills / overwrites those pixels
            k.taps.dy0 = (ph + cv.pad - rf) / 2 + 1; k.taps.dys = -1;
            k.taps.dx0 = (pw + cv.pad - tf) / 2 + 1; k.taps.dxs = -1;
            k.taps.w0 = rf * cv.k + tf; k.taps.wrs = 2 * cv.k; k.taps.wcs = 2;
            if (ncls == 0) {
                q.Hs = k.geo.Hs; q.Ws = k.geo.Ws; q.M = k.geo.M; q.oph = ph; q.opw = pw; q.taps = k.taps;
            } else {
                q.cls[ncls - 1] = k;
            }
            ++ncls;
        }
    if (ncls == 0) return hipSuccess;
    q.ncls = ncls;
    if (ds) {
        if (q.oph != 0 || q.opw != 0 || q.taps.nr != 1 || q.taps.nc != 1 || ds->Co != cv.Co) return hipErrorInvalidValue;
        q.x2 = dzd; q.w2 = c.b16(c.p->arena_off) + ds->dgr_off; q.Kc2 = ds->Co;
    }
    if (sums) {      // the four classes together write every pixel of dx exactly once
        q.bst_z = sums->z; q.bst_mask = sums->mask;
        q.stats = sums->rows; q.stat_rows = VPD_FUSED_ROWS;
        if (accumulate || !vpd_conv_takes_bn_sums(q)) return hipErrorInvalidValue;
    }
    return vpd_launch_conv(q, c.s);
}

hipError_t run_conv_wgrad(const Ctx& c, const ConvInfo& cv, const bf16_t* dz, int dzpad, const bf16_t* x,
                          hipStream_t st, ZeroRanges* collect_zero = nullptr, bool prezeroed = false) {
    WgradParams q;
    memset(&q, 0, sizeof q);
    q.dz = dz; q.dzHp = cv.Hout + 2 * dzpad; q.dzWp = cv.Wout + 2 * dzpad; q.dzC = cv.Co; q.dzpad = dzpad;
    q.x = x;
    if (cv.stem) { q.xHp = c.p->xHp; q.xWp = c.p->xWp; q.xC = 8; }
    else { q.xHp = cv.Hin + 2; q.xWp = cv.Win + 2; q.xC = cv.Ci; }
    q.dw = c.f32(c.p->wg_off) + cv.wg_off;
    float* slab = c.f32(c.p->slab_off);
    q.slab = cv.slab_off >= 0 ? slab + cv.slab_off : (cv.stem ? slab : nullptr);
    q.defer_reduce = 0;
    q.N = c.n; q.Hs = cv.Hout; q.Ws = cv.Wout; q.istr = cv.stride; q.Kc = cv.Kc; q.Co = cv.Co;
    q.M = c.n * cv.Hout * cv.Wout;
    q.taps = conv_taps_fwd(cv);
    q.prefer_halo_1x1 = !c.p->bottleneck;      // BasicBlock students: the three down-sampling 1x1 convs without atomics
    if (cv.slab_off >= 0 && !vpd_wgrad_overwrites(q)) q.slab = nullptr;      // (an A/B switch turned the halo form off: generic kernel)
    if (collect_zero) {                  // dry run at the start of backward: which ranges need zeroing
        if (!vpd_wgrad_overwrites(q) && collect_zero->count < ZR_MAX) {
            collect_zero->ptr[collect_zero->count] = q.dw;
            collect_zero->n4[collect_zero->count++] = (long)cv.ntaps * cv.Co * cv.Kc / 4;
        }
        return hipSuccess;
    }
    if (!vpd_wgrad_overwrites(q) && !prezeroed) {      // the generic kernel accumulates with atomics: zero its range first
        hipError_t e = hipMemsetAsync(q.dw, 0, (size_t)cv.ntaps * cv.Co * cv.Kc * 4, st);
        if (e != hipSuccess) return e;
    }
    if (vpd_wgrad_overwrites(q) && !q.defer_reduce && !cv.stem) {      // time the MFMA kernel alone, then sum its slab
        hipError_t e;
        {
            // class 5 = the grouped per-stage launches (and single stride-1 halo launches); a stride-2 conv's own halo
            // launch (two output tiles, 128 splits) is a different regime: class 6 with the other per-conv launches
            TimeScope ts(c.p, st, cv.stride == 1 ? 5 : 6, conv_flops(cv, c.n));
            q.defer_reduce = 1;
            e = vpd_launch_wgrad(q, st);
        }
        if (e != hipSuccess) return e;
        return vpd_launch_wgrad_reduce(q, st);
    }
    TimeScope ts(c.p, st, cv.stem ? 7 : (vpd_wgrad_overwrites(q) ? 5 : 6), conv_flops(cv, c.n));      // 7: stem kernels
    return vpd_launch_wgrad(q, st);
}

hipError_t run_bn_apply(const Ctx& c, const ConvInfo& cv, int res_kind, const bf16_t* res, const ConvInfo* rcv,
                        bf16_t* out, int relu) {
    BnApplyParams a;
    memset(&a, 0, sizeof a);
    a.z = c.b16(cv.z_off); a.scale = c.bn_scale(cv.bn); a.shift = c.bn_shift(cv.bn);
    a.res_kind = res_kind; a.res = res; a.rHp = cv.Hout + 2; a.rWp = cv.Wout + 2; a.rpad = 1;
    if (rcv) { a.rscale = c.bn_scale(rcv->bn); a.rshift = c.bn_shift(rcv->bn); }
    a.out = out; a.oHp = cv.Hout + 2; a.oWp = cv.Wout + 2; a.opad = 1;
    a.M = c.n * cv.Hout * cv.Wout; a.H = cv.Hout; a.W = cv.Wout; a.C = cv.Co; a.relu = relu;
    return vpd_launch_bn_apply(a, c.s);
}

// train-mode convolution: dense z + per-channel statistics.  Unfused BatchNorm: the statistics go to the SHARED rows,
// which the finalize launch right behind the conv consumes and re-zeroes; fused: to the BatchNorm's own rows.
hipError_t run_conv_train(const Ctx& c, const ConvInfo& cv, const bf16_t* x, float* bn_running) {
    hipError_t e = run_conv_fwd(c, cv, x, c.b16(cv.z_off), 0, true, nullptr, nullptr, nullptr, 0);
    if (e != hipSuccess || c.fused(cv)) return e;
    return run_bn_finalize(c, cv, bn_running);
}

// BatchNorm (+ residual, ReLU) of a train-mode forward: statistics -> normalised padded activation.  rcv: the
// down-sampling branch's conv (res_kind 2), whose BatchNorm is finalized here too.  One launch when fused.
// Pixel tile of the pipelined 3x3 launches next to conv `cv`'s BatchNorm (its own forward / data gradient, and -- same stage, same
// shape -- its neighbours'), when they run in their XCD-affine tile order: the fused BatchNorm launches then take their items in
// the matching block order (bn.hip, vpd_bn_virtual_block).  3x3 stride-1 convolutions with Ci == Co only; 0 otherwise.
int bn_xcd_tile_px(const Ctx& c, const ConvInfo& cv) {
    if (cv.k != 3 || cv.stride != 1 || cv.stem || cv.Ci != cv.Co) return 0;
    const ConvParams q = conv_dgrad_s1_params(c, cv, c.b16(0), c.b16(0), 0);
    return vpd_conv_xcd_tile_px(q);
}

hipError_t run_bn_fwd(const Ctx& c, const ConvInfo& cv, float* bn_running, int res_kind, const bf16_t* res,
                      const ConvInfo* rcv, bf16_t* out, int relu, unsigned char* mask_out = nullptr) {
    if (!c.fused(cv)) return run_bn_apply(c, cv, res_kind, res, rcv, out, relu);      // (finalized by run_conv_train)
    BnApplyParams a;
    memset(&a, 0, sizeof a);
    a.z = c.b16(cv.z_off);
    a.res_kind = res_kind; a.res = res; a.rHp = cv.Hout + 2; a.rWp = cv.Wout + 2; a.rpad = 1;
    a.out = out; a.oHp = cv.Hout + 2; a.oWp = cv.Wout + 2; a.opad = 1;
    a.M = c.n * cv.Hout * cv.Wout; a.H = cv.Hout; a.W = cv.Wout; a.C = cv.Co; a.relu = relu;
    a.mask_out = mask_out;
    a.xcd_tile_px = bn_xcd_tile_px(c, cv);
    BnFusedFwd f;
    memset(&f, 0, sizeof f);
    auto fill = [&](const ConvInfo& k, double** rows, float* count, const float** gamma, const float** beta, float** rm,
                    float** rv, float** mean, float** rstd, float** scale, float** shift) {
        *rows = c.bn_rows(k.bn); *count = (float)(c.n * k.Hout * k.Wout);
        *gamma = c.params + k.bn.w_off; *beta = c.params + k.bn.b_off;
        *rm = bn_running ? bn_running + k.bn.rm_off : nullptr; *rv = bn_running ? bn_running + k.bn.rv_off : nullptr;
        *mean = c.bn_mean(k.bn); *rstd = c.bn_rstd(k.bn); *scale = c.bn_scale(k.bn); *shift = c.bn_shift(k.bn);
    };
    fill(cv, &f.rows, &f.count, &f.gamma, &f.beta, &f.rm, &f.rv, &f.mean, &f.rstd, &f.scale, &f.shift);
    if (rcv) fill(*rcv, &f.rows2, &f.count2, &f.gamma2, &f.beta2, &f.rm2, &f.rv2, &f.mean2, &f.rstd2, &f.scale2, &f.shift2);
    f.momentum = kBnMomentum; f.eps = kBnEps;
    return vpd_launch_bn_fwd_fused(a, f, c.s);
}

// act != null: ReLU mask from the stored activation (needed when a residual was added before the ReLU);
// relu_from_z: mask recomputed as scale*z + shift > 0 (plain conv-BN-ReLU), which saves reading the activation
// mask_bits: the ReLU mask as a bit map (fused launch only; the caller has checked relu_bits_ok): act and write_g are ignored
bool relu_bits_ok(const Ctx& c, const ConvInfo& cv) {
    return c.p->relu_bits && c.fused(cv) && vpd_bn_bwd_fused_ok(c.n * cv.Hout * cv.Wout, cv.Co, false, false);
}
// dy_pooled: dy has not been produced yet -- it is the gradient of the global average pool over cv's output (the last block of
// the network); the fused launch with a ReLU bit map produces it itself, every other path gets the avgpool_bwd launch first
hipError_t run_bn_bwd(const Ctx& c, const ConvInfo& cv, bf16_t* dy, const bf16_t* act, bf16_t* dz, int dzpad,
                      int write_g, float* grads, bool relu_from_z = false, bool reduce_done = false,
                      const unsigned char* mask_bits = nullptr, const float* dy_pooled = nullptr) {
    BnBwdParams b;
    memset(&b, 0, sizeof b);
    b.dy = dy; b.dy_rw = dy; b.z = c.b16(cv.z_off);
    b.act = act; b.aHp = cv.Hout + 2; b.aWp = cv.Wout + 2; b.apad = 1;
    b.mean = c.bn_mean(cv.bn); b.rstd = c.bn_rstd(cv.bn); b.coef = c.bn_coef(cv.bn);
    b.partials = c.stat_rows();
    b.dz = dz; b.dzHp = cv.Hout + 2 * dzpad; b.dzWp = cv.Wout + 2 * dzpad; b.dzpad = dzpad;
    b.M = c.n * cv.Hout * cv.Wout; b.H = cv.Hout; b.W = cv.Wout; b.C = cv.Co; b.write_g = write_g;
    if (mask_bits) { b.mask_bits = mask_bits; b.act = nullptr; b.write_g = 0; write_g = 0; }
    if (relu_from_z && !reduce_done) { b.act = nullptr; b.mscale = c.bn_scale(cv.bn); b.mshift = c.bn_shift(cv.bn); }
    if (reduce_done) b.act = nullptr;        // dy already holds g (masked by the producing dgrad kernel)
    const bool fused = c.fused(cv) && !reduce_done && vpd_bn_bwd_fused_ok(b.M, b.C, b.act != nullptr, write_g != 0);
    if (dy_pooled) {
        static const bool fold = !(getenv("VPD_POOLBWD_FOLD") && !atoi(getenv("VPD_POOLBWD_FOLD")));
        if (fused && mask_bits && fold) { b.dy_pooled = dy_pooled; b.dy_pool_scale = 1.f / (float)(cv.Hout * cv.Wout); }
        else {
            hipError_t e = vpd_launch_avgpool_bwd(dy_pooled, cv.Hout, cv.Wout, cv.Co, c.n, dy, c.s);
            if (e != hipSuccess) return e;
        }
    }
    if (fused) {
        BnFusedBwd f;
        f.rows = c.bn_rows(cv.bn); f.sync = c.ws + cv.bn.sync_off;
        f.err = reinterpret_cast<unsigned*>(c.ws + c.p->syncerr_off);
        f.gamma = c.params + cv.bn.w_off; f.dgamma = grads + cv.bn.w_off; f.dbeta = grads + cv.bn.b_off;
        f.count = (float)b.M;
        return vpd_launch_bn_bwd_fused(b, f, c.s);
    }
    return vpd_launch_bn_bwd(b, (float)b.M, c.params + cv.bn.w_off, grads + cv.bn.w_off, grads + cv.bn.b_off, c.s,
                             reduce_done);
}

// BatchNorm backward whose sums were taken by the producing data gradient (BnSums): finalize + apply
// cvB / dzB: a second BatchNorm fed with the same masked gradient (a down-sampling block's 1x1 branch), same launch
hipError_t run_bn_bwd_apply(const Ctx& c, const ConvInfo& cv, const bf16_t* dy, bf16_t* dz, int dzpad, float* grads,
                            const unsigned char* mask_bits, const ConvInfo* cvB = nullptr, bf16_t* dzB = nullptr) {
    BnBwdParams b;
    memset(&b, 0, sizeof b);
    b.dy = dy; b.z = c.b16(cv.z_off);
    b.mean = c.bn_mean(cv.bn); b.rstd = c.bn_rstd(cv.bn);
    b.dz = dz; b.dzHp = cv.Hout + 2 * dzpad; b.dzWp = cv.Wout + 2 * dzpad; b.dzpad = dzpad;
    b.M = c.n * cv.Hout * cv.Wout; b.H = cv.Hout; b.W = cv.Wout; b.C = cv.Co;
    b.mask_bits = mask_bits;
    b.xcd_tile_px = bn_xcd_tile_px(c, cv);
    BnFusedBwd f;
    memset(&f, 0, sizeof f);
    f.rows = c.bn_rows(cv.bn);
    f.gamma = c.params + cv.bn.w_off; f.dgamma = grads + cv.bn.w_off; f.dbeta = grads + cv.bn.b_off;
    f.count = (float)b.M;
    if (cvB) {
        BnFusedBwd fB;
        memset(&fB, 0, sizeof fB);
        fB.rows = c.bn_rows(cvB->bn);
        fB.gamma = c.params + cvB->bn.w_off; fB.dgamma = grads + cvB->bn.w_off; fB.dbeta = grads + cvB->bn.b_off;
        fB.count = f.count;
        return vpd_launch_bn_bwd_apply_fused(b, f, c.s, &fB, c.b16(cvB->z_off), c.bn_mean(cvB->bn), c.bn_rstd(cvB->bn), dzB);
    }
    return vpd_launch_bn_bwd_apply_fused(b, f, c.s);
}

// ---- a Bottleneck identity block's closing 1x1 convolution together with its BatchNorm, the convolution recomputed instead of
// written and read back (conv_stream.hip, conv1x1_bn_stream_kernel; VPD_BNECK_RECOMPUTE=0: conv + BatchNorm launches) ----
ConvParams conv3_params(const Ctx& c, const ConvInfo& cv, const bf16_t* x) {
    ConvParams q;
    memset(&q, 0, sizeof q);
    q.x = x; q.xHp = cv.Hin + 2; q.xWp = cv.Win + 2; q.xC = cv.Ci;
    q.w = c.b16(c.p->arena_off) + cv.fwd_off;
    q.yHp = cv.Hout; q.yWp = cv.Wout; q.yC = cv.Co; q.ypad = 0;
    q.N = c.n; q.Hs = cv.Hout; q.Ws = cv.Wout; q.osub = 1; q.istr = cv.stride;
    q.Kc = cv.Kc; q.Co = cv.Co; q.M = c.n * cv.Hout * cv.Wout;
    q.taps = conv_taps_fwd(cv);
    return q;
}
bool bneck_recompute_ok(const Ctx& c, const BlockInfo& B) {
    if (!c.p->bottleneck || B.ds || !c.p->train || !c.fused(B.c3) || !relu_bits_ok(c, B.c3)) return false;
    if (B.c3.k != 1 || B.c3.stride != 1) return false;
    return vpd_conv1x1_bn_eligible(conv3_params(c, B.c3, c.b16(B.a2_off)));
}
// ... and a DOWN-SAMPLING block whose closing 1x1 conv and 1x1 branch both have 64 input channels and stride 1 (layer1's first
// block): both convolutions and both BatchNorms in the same launches (conv1x1_bn2_stream_kernel)
ConvParams conv3d_params(const Ctx& c, const BlockInfo& B, const bf16_t* xin) {
    ConvParams q = conv3_params(c, B.c3, c.b16(B.a2_off));
    q.x2 = xin; q.w2 = c.b16(c.p->arena_off) + B.cd.fwd_off; q.Kc2 = B.cd.Kc;
    return q;
}
bool bneck_recompute2_ok(const Ctx& c, const BlockInfo& B) {
    if (!c.p->bottleneck || !B.ds || !c.p->train || !c.fused(B.c3) || !c.fused(B.cd) || !relu_bits_ok(c, B.c3)) return false;
    if (B.c3.k != 1 || B.cd.k != 1 || B.c3.stride != 1 || B.cd.stride != 1 || B.c3.Ci != B.cd.Ci || B.c3.Co != B.cd.Co) return false;
    if (B.c3.Hin != B.cd.Hin || B.c3.Win != B.cd.Win) return false;
    return vpd_conv1x1_bn2_eligible(conv3d_params(c, B, c.b16(B.a2_off)));
}
void fill_bn_fwd(const Ctx& c, const ConvInfo& k, float* bn_running, int M, double** rows, float* count, const float** gamma,
                 const float** beta, float** rm, float** rv, float** mean, float** rstd, float** scale, float** shift) {
    *rows = c.bn_rows(k.bn); *count = (float)M;
    *gamma = c.params + k.bn.w_off; *beta = c.params + k.bn.b_off;
    *rm = bn_running ? bn_running + k.bn.rm_off : nullptr; *rv = bn_running ? bn_running + k.bn.rv_off : nullptr;
    *mean = c.bn_mean(k.bn); *rstd = c.bn_rstd(k.bn); *scale = c.bn_scale(k.bn); *shift = c.bn_shift(k.bn);
}
hipError_t run_conv3d_bn_fwd(const Ctx& c, const BlockInfo& B, const bf16_t* xin, bf16_t* out, unsigned char* mask_out,
                             float* bn_running) {
    hipError_t e;
    for (int k = 0; k < 2; ++k) {      // the two statistics passes
        const ConvInfo& cv = k ? B.cd : B.c3;
        ConvParams q = conv3_params(c, cv, k ? xin : c.b16(B.a2_off));
        q.stats = c.bn_rows(cv.bn); q.stat_rows = VPD_FUSED_ROWS;
        TimeScope ts(c.p, c.s, 4, 0.0);      // (a recomputation: its time counts, its FLOPs are not algorithmic work)
        e = vpd_launch_conv1x1_bn(q, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, c.s);
        if (e != hipSuccess) return e;
    }
    ConvParams q = conv3d_params(c, B, xin);
    q.y = out; q.yHp = B.c3.Hout + 2; q.yWp = B.c3.Wout + 2; q.ypad = 1;
    BnFusedFwd f;
    memset(&f, 0, sizeof f);
    fill_bn_fwd(c, B.c3, bn_running, q.M, &f.rows, &f.count, &f.gamma, &f.beta, &f.rm, &f.rv, &f.mean, &f.rstd, &f.scale, &f.shift);
    fill_bn_fwd(c, B.cd, bn_running, q.M, &f.rows2, &f.count2, &f.gamma2, &f.beta2, &f.rm2, &f.rv2, &f.mean2, &f.rstd2, &f.scale2, &f.shift2);
    f.momentum = kBnMomentum; f.eps = kBnEps;
    TimeScope ts(c.p, c.s, 4, conv_flops(B.c3, c.n) + conv_flops(B.cd, c.n));
    return vpd_launch_conv1x1_bn2(q, &f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mask_out, nullptr, nullptr, 0, 1, c.s);
}
hipError_t run_conv3d_bn_bwd(const Ctx& c, const BlockInfo& B, const bf16_t* xin, bf16_t* dout, const unsigned char* mask_bits,
                             bf16_t* dz3, bf16_t* dzd, float* grads) {
    ConvParams q = conv3d_params(c, B, xin);
    q.y = dout; q.acc_mask = mask_bits;
    BnFusedBwd f3, fd;
    memset(&f3, 0, sizeof f3);
    memset(&fd, 0, sizeof fd);
    f3.rows = c.bn_rows(B.c3.bn); f3.count = (float)q.M;
    f3.gamma = c.params + B.c3.bn.w_off; f3.dgamma = grads + B.c3.bn.w_off; f3.dbeta = grads + B.c3.bn.b_off;
    fd.rows = c.bn_rows(B.cd.bn); fd.count = (float)q.M;
    fd.gamma = c.params + B.cd.bn.w_off; fd.dgamma = grads + B.cd.bn.w_off; fd.dbeta = grads + B.cd.bn.b_off;
    hipError_t e;
    {
        TimeScope ts(c.p, c.s, 4, 0.0);      // (BatchNorm backwards: no algorithmic matrix FLOPs)
        e = vpd_launch_conv1x1_bn2(q, nullptr, &f3, &fd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 2, c.s);
    }
    if (e != hipSuccess) return e;
    TimeScope ts(c.p, c.s, 4, 0.0);
    return vpd_launch_conv1x1_bn2(q, nullptr, &f3, &fd, c.bn_mean(B.c3.bn), c.bn_rstd(B.c3.bn), c.bn_mean(B.cd.bn), c.bn_rstd(B.cd.bn),
                                  nullptr, dz3, dzd, 1, 3, c.s);
}
// forward: statistics pass, then relu(BatchNorm(conv(x)) + res) -> out (padded) + the ReLU bit map
hipError_t run_conv3_bn_fwd(const Ctx& c, const ConvInfo& cv, const bf16_t* x, const bf16_t* res, bf16_t* out,
                            unsigned char* mask_out, float* bn_running) {
    ConvParams q = conv3_params(c, cv, x);
    q.stats = c.bn_rows(cv.bn); q.stat_rows = VPD_FUSED_ROWS;
    hipError_t e;
    {
        TimeScope ts(c.p, c.s, 4, 0.0);      // (the statistics pass is a recomputation: time counted, FLOPs not)
        e = vpd_launch_conv1x1_bn(q, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, c.s);
    }
    if (e != hipSuccess) return e;
    q.stats = nullptr; q.stat_rows = 0;
    q.y = out; q.yHp = cv.Hout + 2; q.yWp = cv.Wout + 2; q.ypad = 1;
    q.res = res; q.rHp = cv.Hout + 2; q.rWp = cv.Wout + 2; q.rC = cv.Co; q.rpad = 1;
    BnFusedFwd f;
    memset(&f, 0, sizeof f);
    f.rows = c.bn_rows(cv.bn); f.count = (float)q.M;
    f.gamma = c.params + cv.bn.w_off; f.beta = c.params + cv.bn.b_off;
    f.rm = bn_running ? bn_running + cv.bn.rm_off : nullptr; f.rv = bn_running ? bn_running + cv.bn.rv_off : nullptr;
    f.mean = c.bn_mean(cv.bn); f.rstd = c.bn_rstd(cv.bn); f.scale = c.bn_scale(cv.bn); f.shift = c.bn_shift(cv.bn);
    f.momentum = kBnMomentum; f.eps = kBnEps;
    TimeScope ts(c.p, c.s, 4, conv_flops(cv, c.n));
    return vpd_launch_conv1x1_bn(q, &f, nullptr, nullptr, nullptr, mask_out, nullptr, 0, 1, c.s);
}
// backward: the sums of g = dout * mask and g * z, then dz = A g + B z + D -> dz (padded by 1), dgamma, dbeta
hipError_t run_conv3_bn_bwd(const Ctx& c, const ConvInfo& cv, const bf16_t* x, bf16_t* dout, const unsigned char* mask_bits,
                            bf16_t* dz, float* grads) {
    ConvParams q = conv3_params(c, cv, x);
    q.y = dout; q.acc_mask = mask_bits;
    BnFusedBwd f;
    memset(&f, 0, sizeof f);
    f.rows = c.bn_rows(cv.bn); f.count = (float)q.M;
    f.gamma = c.params + cv.bn.w_off; f.dgamma = grads + cv.bn.w_off; f.dbeta = grads + cv.bn.b_off;
    hipError_t e;
    {
        TimeScope ts(c.p, c.s, 4, 0.0);      // (a BatchNorm backward: no algorithmic matrix FLOPs)
        e = vpd_launch_conv1x1_bn(q, nullptr, &f, nullptr, nullptr, nullptr, nullptr, 0, 2, c.s);
    }
    if (e != hipSuccess) return e;
    TimeScope ts(c.p, c.s, 4, 0.0);
    return vpd_launch_conv1x1_bn(q, nullptr, &f, c.bn_mean(cv.bn), c.bn_rstd(cv.bn), nullptr, dz, 1, 3, c.s);
}

#define LCHECK(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return fail(#expr, _e);  \
    } while (0)

// min_n = 0 for the train-step entry points: a data-parallel rank whose shard of a ragged last batch is empty still
// takes part in the step (zero loss, zero gradients, bucket events recorded) so that the collective stays matched
int check_call(const vpd_plan* p, const void* ws, int n, int min_n = 1) {
    if (!p || !ws) return fail("null plan / workspace");
    if (p->bound_ws != ws) return fail("workspace not initialised with vpd_plan_init_workspace");
    if (n < min_n || n > p->max_batch) return fail(min_n ? "batch size outside 1..max_batch" : "batch size outside 0..max_batch");
    return 0;
}

// encoder head shared by eval / train: avgpool + fc (+ motion MLP) (+ loss)
int run_head(const Ctx& c, const bf16_t* last_act, float* emb_out, const float* target, bool need_grad,
             float* loss_step, double* loss_accum) {
    vpd_plan* p = c.p;
    const StageInfo& S = p->stages[3];
    LCHECK(vpd_launch_avgpool(last_act, S.H + 2, S.W + 2, 1, S.H, S.W, p->feat, c.n, c.f32(p->pooled_off), c.s));
    // without the motion head nothing re-reads the embedding (the fc backward uses the pooled features and d(emb)): the fc
    // GEMM writes the caller's buffer directly; with it, the head's first layer and its weight gradient read the workspace copy
    float* emb = (emb_out && !p->motion) ? emb_out : c.f32(p->emb_off);
    LCHECK(vpd_launch_sgemm(c.f32(p->pooled_off), c.params + p->fc.w_off, emb, c.params + p->fc.b_off, c.n, p->D, p->feat,
                            0, 1, 0, c.s));
    if (emb_out && emb != emb_out) LCHECK(hipMemcpyAsync(emb_out, emb, (size_t)c.n * p->D * 4, hipMemcpyDeviceToDevice, c.s));
    if (!target) return 0;
    const float* pred = emb;
    int pd = p->D;
    if (p->motion) {
        LCHECK(vpd_launch_sgemm(emb, c.params + p->dec[0].w_off, c.f32(p->h1_off), c.params + p->dec[0].b_off, c.n, 128,
                                p->D, 0, 1, 1, c.s));
        LCHECK(vpd_launch_sgemm(c.f32(p->h1_off), c.params + p->dec[1].w_off, c.f32(p->h2_off),
                                c.params + p->dec[1].b_off, c.n, 128, 128, 0, 1, 1, c.s));
        LCHECK(vpd_launch_sgemm(c.f32(p->h2_off), c.params + p->dec[2].w_off, c.f32(p->pred_off),
                                c.params + p->dec[2].b_off, c.n, 2 * p->D, 128, 0, 1, 0, c.s));
        pred = c.f32(p->pred_off);
        pd = 2 * p->D;
    }
    LCHECK(vpd_launch_mse(pred, target, (long)c.n * pd, need_grad ? c.f32(p->dpred_off) : nullptr, loss_step,
                          loss_accum, c.s));
    return 0;
}

int run_eval_forward(vpd_plan* p, const float* params, const float* x, int n, float* emb_out, const float* target,
                     float* loss_step, double* loss_accum, char* ws, hipStream_t s) {
    Ctx c{p, ws, s, params, n};
    if (x) LCHECK(vpd_launch_pack_input(x, n, p->c_in, p->H, p->W, c.b16(p->xin_off), p->xHp, p->xWp, 3, 8, s));
    // stem: conv + folded BatchNorm + ReLU + max-pool in ONE launch when the stem kernel takes the shape and there are enough
    // images for its image-per-block walk (VPD_STEM_POOL_FUSED=0: conv, then the pooling launch)
    static const bool fuse_pool = !(getenv("VPD_STEM_POOL_FUSED") && !atoi(getenv("VPD_STEM_POOL_FUSED")));
    bool pooled = false;
    if (fuse_pool && n >= 64)
        LCHECK(run_conv_fwd(c, p->stem, c.b16(p->xin_off), c.b16(p->z0_off), 0, false, c.bn_escale(p->stem.bn),
                            c.bn_eshift(p->stem.bn), nullptr, 1, nullptr, c.b16(p->p0_off), &pooled));
    else
        LCHECK(run_conv_fwd(c, p->stem, c.b16(p->xin_off), c.b16(p->z0_off), 0, false, nullptr, nullptr, nullptr, 0));
    if (!pooled) {
        StemPoolParams sp;
        memset(&sp, 0, sizeof sp);
        sp.z = c.b16(p->z0_off); sp.Hz = p->H0; sp.Wz = p->W0;
        sp.scale = c.bn_escale(p->stem.bn); sp.shift = c.bn_eshift(p->stem.bn);
        sp.out = c.b16(p->p0_off); sp.opad = 1; sp.idx = nullptr; sp.N = n; sp.Ho = p->H1; sp.Wo = p->W1; sp.C = 64;
        LCHECK(vpd_launch_stem_pool(sp, s));
    }
    const bf16_t* cur = c.b16(p->p0_off);
    for (auto& B : p->blocks) {
        bf16_t* a1 = c.b16(B.a1_off);
        bf16_t* outp = c.b16(B.out_off);
        const bool pair = B.ds && conv_pair_ok(c, B.c1, B.cd, false);
        if (pair) {      // the down-sampling 1x1 rides in conv1's launch
            const AltConv alt{&B.cd, c.b16(p->stages[B.stage].idn_off), c.bn_escale(B.cd.bn), c.bn_eshift(B.cd.bn), 0};
            LCHECK(run_conv_fwd(c, B.c1, cur, a1, 1, false, c.bn_escale(B.c1.bn), c.bn_eshift(B.c1.bn), nullptr, 1, &alt));
        } else {
            LCHECK(run_conv_fwd(c, B.c1, cur, a1, 1, false, c.bn_escale(B.c1.bn), c.bn_eshift(B.c1.bn), nullptr, 1));
        }
        if (p->bottleneck) {
            bf16_t* a2 = c.b16(B.a2_off);
            LCHECK(run_conv_fwd(c, B.c2, a1, a2, 1, false, c.bn_escale(B.c2.bn), c.bn_eshift(B.c2.bn), nullptr, 1));
            const bf16_t* idn3 = cur;
            if (B.ds) {
                bf16_t* idb = c.b16(p->stages[B.stage].idn_off);
                LCHECK(run_conv_fwd(c, B.cd, cur, idb, 1, false, c.bn_escale(B.cd.bn), c.bn_eshift(B.cd.bn), nullptr, 0));
                idn3 = idb;
            }
            LCHECK(run_conv_fwd(c, B.c3, a2, outp, 1, false, c.bn_escale(B.c3.bn), c.bn_eshift(B.c3.bn), idn3, 1));
            cur = outp;
            continue;
        }
        const bf16_t* idn = cur;
        if (B.ds) {
            bf16_t* idb = c.b16(p->stages[B.stage].idn_off);
            if (!pair) LCHECK(run_conv_fwd(c, B.cd, cur, idb, 1, false, c.bn_escale(B.cd.bn), c.bn_eshift(B.cd.bn), nullptr, 0));
            idn = idb;
        }
        LCHECK(run_conv_fwd(c, B.c2, a1, outp, 1, false, c.bn_escale(B.c2.bn), c.bn_eshift(B.c2.bn), idn, 1));
        cur = outp;
    }
    return run_head(c, cur, emb_out, target, false, loss_step, loss_accum);
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" int vpd_pack_weights(vpd_plan_t* p, const float* params, const float* bn_running, void* workspace,
                                void* stream) {
    if (!p || !workspace || !params) return fail("null argument");
    if (p->bound_ws != workspace) return fail("workspace not initialised with vpd_plan_init_workspace");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    Ctx c{p, ws, s, params, 1};
    LCHECK(vpd_launch_pack_weights(reinterpret_cast<const PackDesc*>(ws + p->desc_off), (int)p->descs.size(),
                                   reinterpret_cast<const int*>(ws + p->bmap_pack_off), (int)p->bmap_pack.size() / 2,
                                   params, c.b16(p->arena_off), s));
    if (bn_running)
        for (BnInfo* b : p->bns)
            LCHECK(vpd_launch_bn_fold(params + b->w_off, params + b->b_off, bn_running + b->rm_off,
                                      bn_running + b->rv_off, kBnEps, c.bn_escale(*b), c.bn_eshift(*b), b->C, s));
    return 0;
}

extern "C" int vpd_forward_eval(vpd_plan_t* p, const float* params, const float* x, int n, float* emb_out,
                                const float* target, float* loss_step, double* loss_accum, void* workspace,
                                void* stream) {
    if (check_call(p, workspace, n)) return -1;
    return run_eval_forward(p, params, x, n, emb_out, target, loss_step, loss_accum, (char*)workspace,
                            (hipStream_t)stream);
}

extern "C" int vpd_forward_train(vpd_plan_t* p, const float* params, float* bn_running, const float* x,
                                 const float* target, int n, float* emb_out, float* loss_step, double* loss_accum,
                                 void* workspace, void* stream) {
    if (check_call(p, workspace, n, 0)) return -1;
    if (!p->train) return fail("plan was created with train=0");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    if (n == 0) {      // empty shard: no crops, no statistics update, zero loss (the running buffers stay as they are)
        if (loss_step) HCHECK(hipMemsetAsync(loss_step, 0, sizeof(float), s));
        return 0;
    }
    Ctx c{p, ws, s, params, n};
    {
        ZeroRanges z;
        memset(&z, 0, sizeof z);
        z.ptr[0] = c.f32(p->partial_off); z.n4[0] = (long)p->partial_bytes / 16; z.count = 1;      // accumulator rows
        if (p->fused_bn) { z.ptr[1] = c.f32(p->fused_off); z.n4[1] = (long)p->fused_bytes / 16; z.count = 2; }
        LCHECK(vpd_launch_zero_ranges(z, s));
    }
    if (x) LCHECK(vpd_launch_pack_input(x, n, p->c_in, p->H, p->W, c.b16(p->xin_off), p->xHp, p->xWp, 3, 8, s));
    // stem: conv -> batch stats -> BN+ReLU+maxpool
    LCHECK(run_conv_fwd(c, p->stem, c.b16(p->xin_off), c.b16(p->z0_off), 0, true, nullptr, nullptr, nullptr, 0));
    LCHECK(run_bn_finalize(c, p->stem, bn_running));
    {
        StemPoolParams sp;
        memset(&sp, 0, sizeof sp);
        sp.z = c.b16(p->z0_off); sp.Hz = p->H0; sp.Wz = p->W0;
        sp.scale = c.bn_scale(p->stem.bn); sp.shift = c.bn_shift(p->stem.bn);
        sp.out = c.b16(p->p0_off); sp.opad = 1; sp.idx = reinterpret_cast<unsigned char*>(ws + p->idx_off);
        sp.N = n; sp.Ho = p->H1; sp.Wo = p->W1; sp.C = 64;
        LCHECK(vpd_launch_stem_pool(sp, s));
    }
    const bf16_t* cur = c.b16(p->p0_off);
    for (auto& B : p->blocks) {
        bf16_t* a1 = c.b16(B.a1_off);
        bf16_t* outp = c.b16(B.out_off);
        const bool pair = B.ds && conv_pair_ok(c, B.c1, B.cd, true);
        if (pair) {      // conv1 and the down-sampling 1x1 in one launch (both read `cur`; statistics to their own rows)
            const AltConv alt{&B.cd, c.b16(B.cd.z_off), nullptr, nullptr, 0};
            LCHECK(run_conv_fwd(c, B.c1, cur, c.b16(B.c1.z_off), 0, true, nullptr, nullptr, nullptr, 0, &alt));
        } else {
            LCHECK(run_conv_train(c, B.c1, cur, bn_running));
        }
        LCHECK(run_bn_fwd(c, B.c1, bn_running, 0, nullptr, nullptr, a1, 1,
                          B.mask1_off ? reinterpret_cast<unsigned char*>(ws + B.mask1_off) : nullptr));
        if (p->bottleneck) {
            bf16_t* a2 = c.b16(B.a2_off);
            LCHECK(run_conv_train(c, B.c2, a1, bn_running));
            LCHECK(run_bn_fwd(c, B.c2, bn_running, 0, nullptr, nullptr, a2, 1,
                              B.mask2_off ? reinterpret_cast<unsigned char*>(ws + B.mask2_off) : nullptr));
            unsigned char* mb3 = p->relu_bits ? reinterpret_cast<unsigned char*>(ws + B.mask_off) : nullptr;
            if (bneck_recompute_ok(c, B)) {      // conv3 + bn3 + identity + ReLU: z3 is never stored
                LCHECK(run_conv3_bn_fwd(c, B.c3, a2, cur, outp, mb3, bn_running));
                cur = outp;
                continue;
            }
            if (bneck_recompute2_ok(c, B)) {     // ... + the 1x1 branch and its BatchNorm: neither z3 nor zd is stored
                LCHECK(run_conv3d_bn_fwd(c, B, cur, outp, mb3, bn_running));
                cur = outp;
                continue;
            }
            LCHECK(run_conv_train(c, B.c3, a2, bn_running));
            if (B.ds) {
                LCHECK(run_conv_train(c, B.cd, cur, bn_running));
                LCHECK(run_bn_fwd(c, B.c3, bn_running, 2, c.b16(B.cd.z_off), &B.cd, outp, 1, mb3));
            } else {
                LCHECK(run_bn_fwd(c, B.c3, bn_running, 1, cur, nullptr, outp, 1, mb3));
            }
            cur = outp;
            continue;
        }
        LCHECK(run_conv_train(c, B.c2, a1, bn_running));
        unsigned char* mbits = p->relu_bits ? reinterpret_cast<unsigned char*>(ws + B.mask_off) : nullptr;
        if (B.ds) {
            if (!pair) LCHECK(run_conv_train(c, B.cd, cur, bn_running));
            LCHECK(run_bn_fwd(c, B.c2, bn_running, 2, c.b16(B.cd.z_off), &B.cd, outp, 1, mbits));
        } else {
            LCHECK(run_bn_fwd(c, B.c2, bn_running, 1, cur, nullptr, outp, 1, mbits));
        }
        cur = outp;
    }
    return run_head(c, cur, emb_out, target, true, loss_step, loss_accum);
}

extern "C" int vpd_backward(vpd_plan_t* p, const float* params, float* grads, int n, void** bucket_events,
                            void* workspace, void* stream) {
    if (check_call(p, workspace, n, 0)) return -1;
    if (!p->train) return fail("plan was created with train=0");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    if (n == 0) {      // empty shard: the gradient of a sum over no crops is zero; every bucket is "ready" at once
        // (lazy: the reducer sums the scratch ranges, and the optimizer step reads them there afterwards)
        p->grads_in_scratch = p->lazy_next;
        p->lazy_next = false;
        HCHECK(hipMemsetAsync(grads, 0, (size_t)p->nparam_padded * sizeof(float), s));
        if (p->grads_in_scratch) HCHECK(hipMemsetAsync(ws + p->wg_off, 0, (size_t)p->wg_elems * sizeof(float), s));
        for (int b = 0; b < 4; ++b)
            if (bucket_events && bucket_events[b]) HCHECK(hipEventRecord((hipEvent_t)bucket_events[b], s));
        return 0;
    }
    Ctx c{p, ws, s, params, n};
    // one launch zeroes the accumulator rows and every weight-gradient range the atomics kernel will add into
    ZeroRanges zr;
    memset(&zr, 0, sizeof zr);
    zr.ptr[0] = c.f32(p->partial_off); zr.n4[0] = (long)p->partial_bytes / 16; zr.count = 1;
    if (p->fused_bn) { zr.ptr[1] = c.f32(p->fused_off); zr.n4[1] = (long)p->fused_bytes / 16; zr.count = 2; }
    bool prezeroed = true;
    {
        auto dry = [&](const ConvInfo& cv, int dzpad) {
            (void)run_conv_wgrad(c, cv, nullptr, dzpad, nullptr, s, &zr);
        };
        for (auto& B : p->blocks) { dry(B.c1, 1); dry(B.c2, 1); if (p->bottleneck) dry(B.c3, 1); if (B.ds) dry(B.cd, 1); }
        dry(p->stem, 0);
        if (zr.count >= ZR_MAX) {      // too many ranges (Bottleneck nets: 30-100 1x1 convs): zero the whole scratch in one range
            const int k = p->fused_bn ? 2 : 1;
            zr.ptr[k] = c.f32(p->wg_off); zr.n4[k] = (long)(p->wg_elems + 3) / 4; zr.count = k + 1;
        }
    }
    LCHECK(vpd_launch_zero_ranges(zr, s));

    // ---- head ----
    if (p->loss_scale != 1.f)      // (fp16 training: vpd_plan_set_loss_scale)
        LCHECK(vpd_launch_scale(c.f32(p->dpred_off), (long)n * (p->motion ? 2 * p->D : p->D), p->loss_scale, s));
    const float* demb = c.f32(p->dpred_off);
    if (p->motion) {
        const LinInfo* L = p->dec;
        // layer 5: pred = h2 W2^T + b
        LCHECK(vpd_launch_sgemm(c.f32(p->dpred_off), c.f32(p->h2_off), grads + L[2].w_off, nullptr, L[2].out, L[2].in, n, 1, 0, 0, s));
        LCHECK(vpd_launch_colsum(c.f32(p->dpred_off), n, L[2].out, grads + L[2].b_off, s));
        LCHECK(vpd_launch_sgemm(c.f32(p->dpred_off), params + L[2].w_off, c.f32(p->dh2_off), nullptr, n, L[2].in, L[2].out, 0, 0, 0, s));
        LCHECK(vpd_launch_relu_mask(c.f32(p->dh2_off), c.f32(p->h2_off), (long)n * 128, s));
        LCHECK(vpd_launch_sgemm(c.f32(p->dh2_off), c.f32(p->h1_off), grads + L[1].w_off, nullptr, L[1].out, L[1].in, n, 1, 0, 0, s));
        LCHECK(vpd_launch_colsum(c.f32(p->dh2_off), n, L[1].out, grads + L[1].b_off, s));
        LCHECK(vpd_launch_sgemm(c.f32(p->dh2_off), params + L[1].w_off, c.f32(p->dh1_off), nullptr, n, L[1].in, L[1].out, 0, 0, 0, s));
        LCHECK(vpd_launch_relu_mask(c.f32(p->dh1_off), c.f32(p->h1_off), (long)n * 128, s));
        LCHECK(vpd_launch_sgemm(c.f32(p->dh1_off), c.f32(p->emb_off), grads + L[0].w_off, nullptr, L[0].out, L[0].in, n, 1, 0, 0, s));
        LCHECK(vpd_launch_colsum(c.f32(p->dh1_off), n, L[0].out, grads + L[0].b_off, s));
        LCHECK(vpd_launch_sgemm(c.f32(p->dh1_off), params + L[0].w_off, c.f32(p->demb_off), nullptr, n, L[0].in, L[0].out, 0, 0, 0, s));
        demb = c.f32(p->demb_off);
    }
    int gi = 0;      // index of the G buffer holding d(out) of the current block
    bf16_t* G[3] = {c.b16(p->G_off[0]), c.b16(p->G_off[1]), c.b16(p->G_off[2])};
    bool pool_pending = false;      // d(out) of the last block has not been written yet: d(pooled) is what there is
    {
        LCHECK(vpd_launch_sgemm(demb, c.f32(p->pooled_off), grads + p->fc.w_off, nullptr, p->D, p->feat, n, 1, 0, 0, s));
        LCHECK(vpd_launch_colsum(demb, n, p->D, grads + p->fc.b_off, s));
        LCHECK(vpd_launch_sgemm(demb, params + p->fc.w_off, c.f32(p->dpooled_off), nullptr, n, p->feat, p->D, 0, 0, 0, s));
        const StageInfo& S = p->stages[3];
        // BasicBlock students: the last block's BatchNorm backward produces d(out) from d(pooled) itself (run_bn_bwd)
        if (!p->bottleneck && !p->blocks.back().ds) pool_pending = true;
        else LCHECK(vpd_launch_avgpool_bwd(c.f32(p->dpooled_off), S.H, S.W, p->feat, n, G[gi], s));
    }
    // grouped mode: eligible convs are queued and launched together when the stage's backward is done
    // (running weight gradients or their slab sums on a second stream was measured 6 % slower in round 2 and, with the persistent
    //  kernels, 1.4 % slower in round 6; confined to a CU partition 40-50 % slower: profiles/r06_ab_wgrad_overlap.txt,
    //  tools/probe/wg_overlap.patch)
    const bool grouped = p->wg_group;
    struct Pending { const ConvInfo* cv; const bf16_t* dz; const bf16_t* x; };
    std::vector<Pending> pending;
    auto make_q = [&](const Pending& pd) {
        const ConvInfo& cv = *pd.cv;
        WgradParams q;
        memset(&q, 0, sizeof q);
        q.dz = pd.dz; q.dzHp = cv.Hout + 2; q.dzWp = cv.Wout + 2; q.dzC = cv.Co; q.dzpad = 1;
        q.x = pd.x; q.xHp = cv.Hin + 2; q.xWp = cv.Win + 2; q.xC = cv.Ci;
        q.dw = c.f32(p->wg_off) + cv.wg_off;
        q.slab = c.f32(p->gslab_off) + cv.gslab_off;
        q.N = n; q.Hs = cv.Hout; q.Ws = cv.Wout; q.istr = cv.stride; q.Kc = cv.Kc; q.Co = cv.Co;
        q.M = n * cv.Hout * cv.Wout;
        q.taps = conv_taps_fwd(cv);
        if (cv.k == 1 && cv.stride == 1 && cv.Co % 128 != 0 && cv.Ci % 128 == 0) {      // operands swapped, result stored transposed
            q.dz = pd.x; q.dzC = cv.Ci; q.x = pd.dz; q.xC = cv.Co; q.Co = cv.Ci; q.Kc = cv.Co; q.transposed = 1;
        }
        return q;
    };
    // `slot`: the stage whose table / schedule cache the 128 x 64 launch uses
    auto flush_group = [&](int slot) -> hipError_t {
        if (pending.empty()) return hipSuccess;
        hipError_t r = hipSuccess;
        // persistent 128-wide tiles (conv_wgrad128_persistent_kernel) for every conv it takes: one launch per 18 problems
        // (a ResNet-50 stage has up to 19: two balanced launches)
        std::vector<Pending> rest;
        {
            std::vector<WgradParams> elig;
            std::vector<double> fl;
            for (const Pending& pd : pending) {
                const WgradParams q = make_q(pd);
                if (vpd_wgrad128_eligible(q)) { elig.push_back(q); fl.push_back(conv_flops(*pd.cv, n)); }
                else rest.push_back(pd);
            }
            const int total = (int)elig.size();
            const int nl = (total + 17) / 18;
            int at = 0;
            for (int l = 0; l < nl && r == hipSuccess; ++l) {
                const int cnt = (total - at + (nl - l) - 1) / (nl - l);
                double flops = 0.0;
                for (int i = 0; i < cnt; ++i) flops += fl[at + i];
                const int sl = (2 * slot + (l & 1)) & 7;
                if (l >= 2) {      // more than 36 problems (ResNet-101's layer3): the table slots are reused -- new shapes per launch
                    if (p->wg2_cache[sl]) { vpd_wgrad128_cache_free(p->wg2_cache[sl]); p->wg2_cache[sl] = nullptr; }
                }
                if (!p->wg2_cache[sl]) p->wg2_cache[sl] = vpd_wgrad128_cache_new();
                TimeScope ts(p, s, 5, flops);
                r = vpd_launch_wgrad128_group(elig.data() + at, cnt, p->wg2_cache[sl], ws + p->wg2_tbl_off[sl], s);
                at += cnt;
            }
        }
        size_t done = 0;
        while (done < rest.size() && r == hipSuccess) {
            WgradParams qs[12];
            const int cnt = (int)std::min<size_t>(12, rest.size() - done);
            double flops = 0.0;
            // one launch of the 64 x 64 grouped kernel: one halo geometry (stage)
            int take = 0;
            for (int i = 0; i < cnt; ++i) {
                if (i > 0 && rest[done + i].cv->Hout != rest[done].cv->Hout) break;
                qs[take++] = make_q(rest[done + i]);
                flops += conv_flops(*rest[done + i].cv, n);
            }
            {
                TimeScope ts(p, s, 5, flops);
                r = vpd_launch_wgrad_group(qs, take, s);
            }
            done += take;
        }
        pending.clear();
        return r;
    };
    // wgrad of `cv` may start once everything enqueued on the main stream so far (its dz) is done
    auto queue_wgrad = [&](const ConvInfo& cv, const bf16_t* dz, int dzpad, const bf16_t* x) -> hipError_t {
        if (grouped && cv.dz_own_off && dz == c.b16(cv.dz_own_off)) {
            pending.push_back({&cv, dz, x});
            return hipSuccess;
        }
        // (a bucket is handed over -- unpacked, its event recorded -- only behind its stage's flush_group, i.e. behind these sums)
        return run_conv_wgrad(c, cv, dz, dzpad, x, s, nullptr, prezeroed);
    };
    // lazy: the caller asked for it (vpd_plan_set_lazy_grads).  With bucket events the reducer then sums the scratch ranges
    // (vpd_plan_bucket_scratch_range) and the non-conv tensors of the flat buffer instead of the whole flat buffer
    const bool lazy = p->lazy_next;
    p->lazy_next = false;
    p->grads_in_scratch = lazy;
    auto unpack_bucket = [&](int b) -> int {
        int nb = (int)p->bmap_unpack[b].size() / 2;
        if (lazy) nb = b == 3 ? p->nstem_unpack_blocks : 0;      // the stem's row-tap packing is undone here either way
        if (nb > 0)
            LCHECK(vpd_launch_unpack_grads(reinterpret_cast<const PackDesc*>(ws + p->desc_off), (int)p->descs.size(),
                                           reinterpret_cast<const int*>(ws + p->bmap_unpack_off[b]), nb,
                                           c.f32(p->wg_off), grads, s));
        if (bucket_events && bucket_events[b]) LCHECK(hipEventRecord((hipEvent_t)bucket_events[b], s));
        return 0;
    };

    // End of a stage's backward (called after every block): launch the stage's grouped weight gradients and hand its
    // gradient bucket over -- except that layer4's (stage 3) wait for layer3's when wg_merge34: one launch then carries both
    // stages (their tasks fill the chip together where each stage alone leaves CUs idle), and bucket 0 follows it.
    std::vector<int> deferred_buckets;
    auto stage_end = [&](int bi) -> int {
        const BlockInfo& B = p->blocks[bi];
        const bool last_of_stage = bi == 0 || p->blocks[bi - 1].stage != B.stage;
        if (!last_of_stage) return 0;
        const bool defer = grouped && p->wg_merge34 && B.stage == 3 && bi > 0;
        if (defer) { deferred_buckets.push_back(3 - B.stage); return 0; }
        LCHECK(flush_group(B.stage));
        for (int b : deferred_buckets)
            if (unpack_bucket(b)) return -1;
        deferred_buckets.clear();
        if (bi > 0) return unpack_bucket(3 - B.stage);
        return 0;
    };
    // block-output BatchNorm (A: the block's last conv) and the down-sampling branch's BatchNorm (Bc) in one launch: same dy,
    // same ReLU mask (bn_bwd_fused2_kernel); false: not applicable here, the caller runs them one after the other
    auto bn_bwd_pair = [&](const ConvInfo& A, const ConvInfo& Bc, bf16_t* dout_, const bf16_t* out_act, bf16_t* dzA,
                           bf16_t* dzB, bool* done) -> int {
        *done = false;
        if (!(c.fused(A) && c.fused(Bc) && A.Co == Bc.Co && vpd_bn_bwd_fused2_ok(n * A.Hout * A.Wout, A.Co))) return 0;
        BnBwdParams b;
        memset(&b, 0, sizeof b);
        b.dy = dout_; b.dy_rw = dout_; b.z = c.b16(A.z_off);
        b.act = out_act; b.aHp = A.Hout + 2; b.aWp = A.Wout + 2; b.apad = 1;
        b.mean = c.bn_mean(A.bn); b.rstd = c.bn_rstd(A.bn);
        b.dz = dzA; b.dzHp = A.Hout + 2; b.dzWp = A.Wout + 2; b.dzpad = 1;
        b.M = n * A.Hout * A.Wout; b.H = A.Hout; b.W = A.Wout; b.C = A.Co;
        BnFusedBwd fA, fB;
        fA.rows = c.bn_rows(A.bn); fA.sync = c.ws + A.bn.sync_off;
        fA.err = reinterpret_cast<unsigned*>(c.ws + p->syncerr_off);
        fA.gamma = params + A.bn.w_off; fA.dgamma = grads + A.bn.w_off; fA.dbeta = grads + A.bn.b_off;
        fA.count = (float)b.M;
        fB = fA;
        fB.rows = c.bn_rows(Bc.bn);
        fB.gamma = params + Bc.bn.w_off; fB.dgamma = grads + Bc.bn.w_off; fB.dbeta = grads + Bc.bn.b_off;
        LCHECK(vpd_launch_bn_bwd_fused2(b, fA, fB, c.b16(Bc.z_off), c.bn_mean(Bc.bn), c.bn_rstd(Bc.bn), dzB, s));
        *done = true;
        return 0;
    };
    std::vector<char> bn2_sums_for(p->blocks.size(), 0);       // ... its sums taken by the next block's dgrad (BnSums)
    static const bool pair_sums = !(getenv("VPD_DGRAD_SUMS_PAIR") && !atoi(getenv("VPD_DGRAD_SUMS_PAIR")));
    for (int bi = (int)p->blocks.size() - 1; bi >= 0; --bi) {
        BlockInfo& B = p->blocks[bi];
        const StageInfo& S = p->stages[B.stage];
        const int par = bi & 1;
        const bf16_t* xin = bi == 0 ? c.b16(p->p0_off) : c.b16(p->blocks[bi - 1].out_off);
        bf16_t* dout = G[gi];
        bf16_t* da1 = G[(gi + 1) % 3];
        bf16_t* dnew = G[(gi + 2) % 3];
        bf16_t* dz2 = c.b16(grouped && B.c2.dz_own_off ? B.c2.dz_own_off : S.dz2_off[par]);
        bf16_t* dz1 = c.b16(grouped && B.c1.dz_own_off ? B.c1.dz_own_off : S.dz1_off[par]);
        if (pool_pending && (p->bottleneck || B.ds || bn2_sums_for[bi])) {      // (not the path that produces d(out) itself)
            LCHECK(vpd_launch_avgpool_bwd(c.f32(p->dpooled_off), S.H, S.W, p->feat, n, dout, s));
            pool_pending = false;
        }
        if (p->bottleneck) {
            bf16_t* dz3 = c.b16(grouped && B.c3.dz_own_off ? B.c3.dz_own_off : S.dz3_off);
            bf16_t* da2 = c.b16(p->T_off[0]);
            bf16_t* da1b = c.b16(p->T_off[1]);
            // bn3 (+ReLU of the block output); leaves g = dout*[out>0] in dout -- or, for identity blocks with the ReLU bit map,
            // leaves dout alone: conv1's data gradient masks it when it adds the identity path (as in the BasicBlock path)
            const unsigned char* mb3 = (!B.ds && relu_bits_ok(c, B.c3)) ? reinterpret_cast<const unsigned char*>(ws + B.mask_off) : nullptr;
            bool bn3_pair = false;      // down-sampling block: bn3 and the 1x1 branch's BatchNorm in one launch
            if (B.ds && bneck_recompute2_ok(c, B)) {
                LCHECK(run_conv3d_bn_bwd(c, B, xin, dout, reinterpret_cast<const unsigned char*>(ws + B.mask_off), dz3,
                                         c.b16(grouped && B.cd.dz_own_off ? B.cd.dz_own_off : S.dzd_off), grads));
                bn3_pair = true;
            } else if (B.ds)
                if (bn_bwd_pair(B.c3, B.cd, dout, c.b16(B.out_off), dz3,
                                c.b16(grouped && B.cd.dz_own_off ? B.cd.dz_own_off : S.dzd_off), &bn3_pair)) return -1;
            if (bneck_recompute_ok(c, B)) LCHECK(run_conv3_bn_bwd(c, B.c3, c.b16(B.a2_off), dout, mb3, dz3, grads));
            else if (!bn3_pair) LCHECK(run_bn_bwd(c, B.c3, dout, c.b16(B.out_off), dz3, 1, 1, grads, false, false, mb3));
            LCHECK(queue_wgrad(B.c3, dz3, 1, c.b16(B.a2_off)));
            // layer3 / layer4 (vpd_plan_create, dgrad_sums): the sums of bn2 / bn1 ride in the data gradients that produce their dy
            if (B.mask2_off && dgrad_takes_sums(c, B.c3, 0)) {
                const unsigned char* m2 = reinterpret_cast<const unsigned char*>(ws + B.mask2_off);
                const BnSums sm{c.b16(B.c2.z_off), m2, c.bn_rows(B.c2.bn), nullptr, nullptr};
                LCHECK(run_conv_dgrad(c, B.c3, dz3, da2, 0, nullptr, nullptr, nullptr, &sm));
                LCHECK(run_bn_bwd_apply(c, B.c2, da2, dz2, 1, grads, m2));
            } else {
                LCHECK(run_conv_dgrad(c, B.c3, dz3, da2, 0));
                LCHECK(run_bn_bwd(c, B.c2, da2, nullptr, dz2, 1, 0, grads, true));
            }
            LCHECK(queue_wgrad(B.c2, dz2, 1, c.b16(B.a1_off)));
            if (B.mask1_off && dgrad_takes_sums(c, B.c2, 0)) {
                const unsigned char* m1 = reinterpret_cast<const unsigned char*>(ws + B.mask1_off);
                const BnSums sm{c.b16(B.c1.z_off), m1, c.bn_rows(B.c1.bn), nullptr, nullptr};
                LCHECK(run_conv_dgrad(c, B.c2, dz2, da1b, 0, nullptr, nullptr, nullptr, &sm));
                LCHECK(run_bn_bwd_apply(c, B.c1, da1b, dz1, 1, grads, m1));
            } else {
                LCHECK(run_conv_dgrad(c, B.c2, dz2, da1b, 0));
                LCHECK(run_bn_bwd(c, B.c1, da1b, nullptr, dz1, 1, 0, grads, true));
            }
            LCHECK(queue_wgrad(B.c1, dz1, 1, xin));
            if (B.ds) {
                bf16_t* dzd = c.b16(grouped && B.cd.dz_own_off ? B.cd.dz_own_off : S.dzd_off);
                if (!bn3_pair) LCHECK(run_bn_bwd(c, B.cd, dout, nullptr, dzd, 1, 0, grads));
                LCHECK(queue_wgrad(B.cd, dzd, 1, xin));
                LCHECK(run_conv_dgrad(c, B.c1, dz1, dnew, 0));      // 1x1 stride 1: writes every input pixel
                LCHECK(run_conv_dgrad(c, B.cd, dzd, dnew, 1));      // adds onto the pixels the strided 1x1 reads
                gi = (gi + 2) % 3;
            } else {
                // identity path + conv path = d(out) of the previous block
                LCHECK(run_conv_dgrad(c, B.c1, dz1, dout, 1, nullptr, nullptr, mb3));
            }
            if (stage_end(bi)) return -1;
            continue;
        }
        // bn2 (+ReLU of the block output); leaves g = dout*[out>0] in dout.  Already done when the NEXT block's conv1
        // data gradient (the previous iteration of this loop) carried it in its epilogue.
        bool bn_pair = false;      // conv2's BatchNorm and the 1x1 branch's BatchNorm in one launch (same dy, same ReLU mask)
        if (B.ds && bn2_sums_for[bi]) {      // both sums were taken by the next block's data gradient: one finalize + apply launch for both
            const unsigned char* mb = reinterpret_cast<const unsigned char*>(ws + B.mask_off);
            LCHECK(run_bn_bwd_apply(c, B.c2, dout, dz2, 1, grads, mb, &B.cd,
                                    c.b16(grouped && B.cd.dz_own_off ? B.cd.dz_own_off : S.dzd_off)));
            bn_pair = true;
        } else if (B.ds)
            if (bn_bwd_pair(B.c2, B.cd, dout, c.b16(B.out_off), dz2,
                            c.b16(grouped && B.cd.dz_own_off ? B.cd.dz_own_off : S.dzd_off), &bn_pair)) return -1;
        // plain (identity) blocks: ReLU mask from the forward's bit map; g = dout * mask is neither written back nor re-read --
        // conv1's data gradient, which adds the identity path, masks dout itself (ConvParams::acc_mask)
        const unsigned char* mbits = nullptr;
        if (!B.ds && relu_bits_ok(c, B.c2))
            mbits = reinterpret_cast<const unsigned char*>(ws + B.mask_off);
        if (bn2_sums_for[bi] && !B.ds)      // (the next block's conv1 data gradient took the sums: mbits is set, dout is left alone)
            LCHECK(run_bn_bwd_apply(c, B.c2, dout, dz2, 1, grads, mbits));
        else if (bn2_sums_for[bi]) { /* down-sampling block: applied above */ }
        else if (!bn_pair) {
            LCHECK(run_bn_bwd(c, B.c2, dout, c.b16(B.out_off), dz2, 1, 1, grads, false, false, mbits,
                              pool_pending ? c.f32(p->dpooled_off) : nullptr));
            pool_pending = false;
        }
        LCHECK(queue_wgrad(B.c2, dz2, 1, c.b16(B.a1_off)));
        if (B.mask1_off && dgrad_takes_sums(c, B.c2, 0)) {
            // bn1's sums ride in conv2's data gradient; its BatchNorm launch only finalizes and applies
            const unsigned char* m1 = reinterpret_cast<const unsigned char*>(ws + B.mask1_off);
            const BnSums sm{c.b16(B.c1.z_off), m1, c.bn_rows(B.c1.bn), nullptr, nullptr};
            LCHECK(run_conv_dgrad(c, B.c2, dz2, da1, 0, nullptr, nullptr, nullptr, &sm));
            LCHECK(run_bn_bwd_apply(c, B.c1, da1, dz1, 1, grads, m1));
        } else {
            LCHECK(run_conv_dgrad(c, B.c2, dz2, da1, 0));
            LCHECK(run_bn_bwd(c, B.c1, da1, nullptr, dz1, 1, 0, grads, true));
        }
        bf16_t* const dzd_pre = B.ds ? c.b16(grouped && B.cd.dz_own_off ? B.cd.dz_own_off : S.dzd_off) : nullptr;
        LCHECK(queue_wgrad(B.c1, dz1, 1, xin));
        if (B.ds) {
            bf16_t* dzd = dzd_pre;      // (its own buffer when it joins the stage's launch)
            if (!bn_pair) LCHECK(run_bn_bwd(c, B.cd, dout, nullptr, dzd, 1, 0, grads));
            LCHECK(queue_wgrad(B.cd, dzd, 1, xin));
            if (conv_pair_ok(c, B.c1, B.cd, true)) {
                // one launch: the 1x1 branch's data gradient is extra K-steps of the even-even class.  Its result is d(out) of
                // the previous stage's last block: the sums of that block's bn2 are taken here
                const BnSums* smp = nullptr;
                BnSums sm;
                static const bool s2sums = !(getenv("VPD_DGRAD_SUMS_S2") && !atoi(getenv("VPD_DGRAD_SUMS_S2")));
                if (bi > 0 && p->dgrad_sums && s2sums && (B.c1.Hin % 2) == 0 && (B.c1.Win % 2) == 0) {
                    const BlockInfo& Bp = p->blocks[bi - 1];
                    if (!Bp.ds && relu_bits_ok(c, Bp.c2)) {
                        sm = BnSums{c.b16(Bp.c2.z_off), reinterpret_cast<const unsigned char*>(ws + Bp.mask_off), c.bn_rows(Bp.c2.bn), nullptr, nullptr};
                        smp = &sm;
                        bn2_sums_for[bi - 1] = true;
                    }
                }
                LCHECK(run_conv_dgrad(c, B.c1, dz1, dnew, 0, &B.cd, dzd, nullptr, smp));
            } else {
                LCHECK(run_conv_dgrad(c, B.c1, dz1, dnew, 0));      // writes every input pixel (3x3 covers all classes)
                LCHECK(run_conv_dgrad(c, B.cd, dzd, dnew, 1));      // adds onto the even-even pixels
            }
            gi = (gi + 2) % 3;
        } else {
            // dout holds g (or, with the bit map, d(out) and the mask is applied here): identity path + conv path.
            // The result is d(out) of the previous block: when that block is a plain one too, the sums of its bn2 are taken here
            const BnSums* smp = nullptr;
            BnSums sm;
            if (bi > 0 && mbits) {
                const BlockInfo& Bp = p->blocks[bi - 1];
                if (!Bp.ds && Bp.stage == B.stage && relu_bits_ok(c, Bp.c2) &&
                    dgrad_takes_sums(c, B.c1, 1)) {
                    sm = BnSums{c.b16(Bp.c2.z_off), reinterpret_cast<const unsigned char*>(ws + Bp.mask_off), c.bn_rows(Bp.c2.bn), nullptr, nullptr};
                    smp = &sm;
                    bn2_sums_for[bi - 1] = true;
                } else if (Bp.ds && Bp.stage == B.stage && pair_sums && c.fused(Bp.c2) && c.fused(Bp.cd) &&
                           Bp.c2.Co == Bp.cd.Co && relu_bits_ok(c, Bp.c2) && dgrad_takes_sums(c, B.c1, 1, true)) {
                    // a down-sampling block: conv2's BatchNorm and the 1x1 branch's see the same g -- both sums here
                    sm = BnSums{c.b16(Bp.c2.z_off), reinterpret_cast<const unsigned char*>(ws + Bp.mask_off), c.bn_rows(Bp.c2.bn),
                                c.b16(Bp.cd.z_off), c.bn_rows(Bp.cd.bn)};
                    smp = &sm;
                    bn2_sums_for[bi - 1] = true;
                }
            }
            LCHECK(run_conv_dgrad(c, B.c1, dz1, dout, 1, nullptr, nullptr, mbits, smp));
        }
        if (stage_end(bi)) return -1;
    }
    // ---- stem ----
    {
        StemPoolBwdParams sb;
        memset(&sb, 0, sizeof sb);
        sb.dpool = G[gi]; sb.idx = reinterpret_cast<const unsigned char*>(ws + p->idx_off); sb.z = c.b16(p->z0_off);
        sb.mean = c.bn_mean(p->stem.bn); sb.rstd = c.bn_rstd(p->stem.bn);
        sb.scale = c.bn_scale(p->stem.bn); sb.shift = c.bn_shift(p->stem.bn);
        sb.g = c.b16(p->g0_off); sb.partials = c.stat_rows();
        sb.pooled = c.b16(p->p0_off); sb.ppad = 1;
        sb.gamma_p = params + p->stem.bn.w_off; sb.beta_p = params + p->stem.bn.b_off;
        sb.M = n * p->H0 * p->W0; sb.Hz = p->H0; sb.Wz = p->W0; sb.Ho = p->H1; sb.Wo = p->W1; sb.C = 64;
        LCHECK(vpd_launch_stem_pool_bwd(sb, (float)sb.M, params + p->stem.bn.w_off, grads + p->stem.bn.w_off,
                                        grads + p->stem.bn.b_off, c.bn_coef(p->stem.bn), c.b16(p->dz0_off), s));
        LCHECK(queue_wgrad(p->stem, c.b16(p->dz0_off), 0, c.b16(p->xin_off)));
    }
    return unpack_bucket(3);
}

extern "C" int vpd_augment_crops(const unsigned char* rgb_u8, const unsigned char* flow_u8, const unsigned char* mask_u8,
                                 const float* noise, const vpd_aug_params* params, int n, int height, int width,
                                 int out_dim, const float* mean_std6, float noise_sd, float* out_nchw, float* scratch,
                                 void* stream) {
    if (!rgb_u8 || !params || !mean_std6 || !out_nchw || !scratch) return fail("null argument");
    if (n < 1 || height < 1 || width < 1 || out_dim < 1) return fail("bad shape");
    LCHECK(vpd_launch_augment(rgb_u8, flow_u8, mask_u8, noise, params, n, height, width, out_dim, mean_std6, noise_sd,
                              out_nchw, nullptr, 0, 0, 0, scratch, (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_plan_stage_crops(vpd_plan_t* p, const unsigned char* rgb_u8, const unsigned char* flow_u8,
                                    const unsigned char* mask_u8, const float* noise, const vpd_aug_params* params,
                                    int n, int height, int width, const float* mean_std6, float noise_sd,
                                    float* scratch, void* workspace, void* stream) {
    if (n > 65535) return fail("more than 65535 crops per staging call");      // (aug_apply_kernel: one grid row per crop)
    if (check_call(p, workspace, n)) return -1;
    if (!rgb_u8 || !params || !mean_std6 || !scratch) return fail("null argument");
    if (p->H != p->W) return fail("the input pipeline resizes to a square img_dim");
    if ((p->c_in == 5) != (flow_u8 != nullptr)) return fail("flow_u8 must be given exactly when the plan has 5 input channels");
    char* ws = (char*)workspace;
    LCHECK(vpd_launch_augment(rgb_u8, flow_u8, mask_u8, noise, params, n, height, width, p->H, mean_std6, noise_sd,
                              nullptr, reinterpret_cast<bf16_t*>(ws + p->xin_off), p->xHp, p->xWp, 3, scratch,
                              (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_plan_stage_views(vpd_plan_t* p, const unsigned char* rgb_u8, const unsigned char* flow_u8, int n_frames,
                                    int k_views, int height, int width, const float* mean_std6, void* workspace,
                                    void* stream) {
    if (n_frames < 0 || (k_views != 1 && k_views != 2)) return fail("k_views must be 1 (frame) or 2 (frame, h-flip)");
    // (aug_views_kernel: one grid row per view; a launch fails opaquely beyond gridDim.y = 65535)
    if ((long)n_frames * k_views > 65535) return fail("n_frames * k_views exceeds 65535 views per staging call");
    if (check_call(p, workspace, n_frames * k_views)) return -1;
    if (!rgb_u8 || !mean_std6) return fail("null argument");
    if (height != p->H || width != p->W) return fail("inference views are not resized: the frames must have the plan's size");
    if (width % 4) return fail("width must be a multiple of 4");
    if ((p->c_in == 5) != (flow_u8 != nullptr)) return fail("flow_u8 must be given exactly when the plan has 5 input channels");
    if (p->c_in != 3 && p->c_in != 5) return fail("inference views: 3 or 5 input channels");
    if (n_frames == 0) return 0;
    char* ws = (char*)workspace;
    LCHECK(vpd_launch_views(rgb_u8, flow_u8, n_frames, k_views, height, width, mean_std6,
                            reinterpret_cast<bf16_t*>(ws + p->xin_off), p->xHp, p->xWp, 3, (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_adamw_step(float* params, const float* grads, float* adam_m, float* adam_v, long long numel,
                              double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                              void* stream) {
    if (!params || !grads || !adam_m || !adam_v) return fail("null argument");
    if (numel % 4) return fail("numel must be a multiple of 4 (use vpd_plan_param_numel)");
    if (step < 1) return fail("step is 1-based");
    LCHECK(vpd_launch_adamw(params, grads, adam_m, adam_v, (long)numel, lr, beta1, beta2, eps, weight_decay, step,
                            (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_plan_adamw_step(vpd_plan_t* p, float* params, const float* grads, float* adam_m, float* adam_v,
                                   long long numel, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int step, void* workspace, void* stream) {
    if (!p || !params || !grads || !adam_m || !adam_v || !workspace) return fail("null argument");
    if (p->bound_ws != workspace) return fail("workspace not initialised with vpd_plan_init_workspace");
    if (numel % 4 || numel < p->nparam_padded) return fail("numel must be a multiple of 4 and cover the plan's parameters");
    if (step < 1) return fail("step is 1-based");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    LCHECK(vpd_launch_adamw_pack(reinterpret_cast<const PackDesc*>(ws + p->desc_off),
                                 reinterpret_cast<const int*>(ws + p->bmap_adam_off), (int)p->bmap_adam.size() / 2, params,
                                 grads, adam_m, adam_v, reinterpret_cast<bf16_t*>(ws + p->arena_off), lr, beta1, beta2,
                                 eps, weight_decay, step, s,
                                 p->grads_in_scratch ? reinterpret_cast<const float*>(ws + p->wg_off) : nullptr,
                                 1.0f / p->loss_scale));
    p->grads_in_scratch = false;      // consumed (the scratch is rewritten by the next backward)
    LCHECK(vpd_launch_pack_weights(reinterpret_cast<const PackDesc*>(ws + p->desc_off), (int)p->descs.size(),
                                   reinterpret_cast<const int*>(ws + p->bmap_pack_off), p->nstem_pack_blocks, params,
                                   reinterpret_cast<bf16_t*>(ws + p->arena_off), s));
    if (numel > p->nparam_padded)      // tensors the plan does not use (a motion head on a plan built without it)
        LCHECK(vpd_launch_adamw(params + p->nparam_padded, grads + p->nparam_padded, adam_m + p->nparam_padded,
                                adam_v + p->nparam_padded, (long)(numel - p->nparam_padded), lr, beta1, beta2, eps,
                                weight_decay, step, s, 1.0f / p->loss_scale));
    return 0;
}

// Loss scale of the NEXT backward passes of this plan and of vpd_plan_adamw_step (fp16 training; the reference's GradScaler,
// models/util.py:55-57, train_vpd_model.py:105): vpd_backward multiplies d(loss)/d(pred) by `scale`, so every gradient it leaves --
// flat buffer, weight-gradient scratch, what a reducer sums -- is scale x its value (no fp16 activation gradient underflows), and
// vpd_plan_adamw_step multiplies the gradients it reads by 1 / scale.  1 (the default) is exact arithmetic: nothing changes.
extern "C" int vpd_plan_set_loss_scale(vpd_plan_t* p, float scale) {
    if (!p) return fail("null plan");
    if (!(scale > 0.f) || !(scale < 3.0e38f)) return fail("loss scale must be a positive finite number");
    p->loss_scale = scale;
    return 0;
}

extern "C" int vpd_plan_set_lazy_grads(vpd_plan_t* p, int on) {
    if (!p) return fail("null plan");
    p->lazy_next = on != 0;
    return 0;
}
extern "C" int vpd_plan_grads_pending(const vpd_plan_t* p) { return p && p->grads_in_scratch ? 1 : 0; }
extern "C" int vpd_plan_materialize_grads(vpd_plan_t* p, float* grads, void* workspace, void* stream) {
    if (!p || !grads || !workspace) return fail("null argument");
    if (p->bound_ws != workspace) return fail("workspace not initialised with vpd_plan_init_workspace");
    if (!p->grads_in_scratch) return 0;
    char* ws = (char*)workspace;
    for (int b = 0; b < 4; ++b) {
        const int nb = (int)p->bmap_unpack[b].size() / 2;
        const int skip = b == 3 ? p->nstem_unpack_blocks : 0;      // the stem was unpacked by the backward itself
        if (nb > skip)
            LCHECK(vpd_launch_unpack_grads(reinterpret_cast<const PackDesc*>(ws + p->desc_off), (int)p->descs.size(),
                                           reinterpret_cast<const int*>(ws + p->bmap_unpack_off[b]) + 2 * skip, nb - skip,
                                           reinterpret_cast<const float*>(ws + p->wg_off), grads, (hipStream_t)stream));
    }
    p->grads_in_scratch = false;
    return 0;
}

extern "C" int vpd_plan_sync_errors(vpd_plan_t* p, void* workspace, void* stream, unsigned* count_out) {
    if (!p || !workspace || !count_out) return fail("null argument");
    if (p->bound_ws != workspace) return fail("workspace not initialised with vpd_plan_init_workspace");
    hipStream_t s = (hipStream_t)stream;
    unsigned v = 0;
    HCHECK(hipMemcpyAsync(&v, (char*)workspace + p->syncerr_off, sizeof v, hipMemcpyDeviceToHost, s));
    HCHECK(hipStreamSynchronize(s));
    *count_out = v;
    return 0;
}

extern "C" int vpd_plan_set_timing(vpd_plan_t* p, int enable) {
    if (!p) return fail("null plan");
    p->timing = enable != 0;
    return 0;
}

// Sums (and clears) the recorded launches: out[4*cls + {0,1,2}] = {launches, milliseconds, flops}
extern "C" int vpd_plan_read_timing(vpd_plan_t* p, double* out, int nclasses) {
    if (!p || !out || nclasses < 8) return fail("bad argument");
    for (int i = 0; i < 3 * nclasses; ++i) out[i] = 0.0;
    for (auto& t : p->timed) {
        HCHECK(hipEventSynchronize(t.b));
        float ms = 0.f;
        HCHECK(hipEventElapsedTime(&ms, t.a, t.b));
        const int cls = t.cls;
        p->ev_pool.push_back(t.a); p->ev_pool.push_back(t.b);
        if (cls < 0 || cls >= nclasses) continue;
        out[3 * cls + 0] += 1.0; out[3 * cls + 1] += ms; out[3 * cls + 2] += t.flops;
    }
    p->timed.clear();
    return 0;
}

extern "C" int vpd_graph_capture_eval(vpd_plan_t* p, const float* params, const float* x, int n, float* emb_out,
                                      void* workspace, void* stream) {
    if (check_call(p, workspace, n)) return -1;
    hipStream_t s = (hipStream_t)stream;
    for (size_t i = 0; i < p->graphs.size(); ++i)
        if (p->graphs[i].n == n) {
            (void)hipGraphExecDestroy(p->graphs[i].e);
            (void)hipGraphDestroy(p->graphs[i].g);
            p->graphs.erase(p->graphs.begin() + i);
            break;
        }
    HCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = run_eval_forward(p, params, x, n, emb_out, nullptr, nullptr, nullptr, (char*)workspace, s);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return -1; }
    if (e != hipSuccess) return fail("hipStreamEndCapture", e);
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); return fail("hipGraphInstantiate", e); }
    p->graphs.push_back({n, g, ge});
    return 0;
}

extern "C" int vpd_graph_launch_eval(vpd_plan_t* p, int n, void* stream) {
    for (auto& g : p->graphs)
        if (g.n == n) {
            HCHECK(hipGraphLaunch(g.e, (hipStream_t)stream));
            return 0;
        }
    return fail("no captured eval graph for this batch size");
}

// ---------------------------------------------------------------------------
// single-operator entry points for the parity tests
// ---------------------------------------------------------------------------
extern "C" int vpd_op_conv_bm(int M, int Co) { return vpd_conv_bm(M, Co); }
extern "C" size_t vpd_op_wgrad_slab_bytes(void) { return vpd_wgrad_slab_bytes(); }

static TapSet tapset_from(const int* t) {
    TapSet ts;
    ts.nr = t[0]; ts.nc = t[1]; ts.dy0 = t[2]; ts.dys = t[3]; ts.dx0 = t[4]; ts.dxs = t[5];
    ts.w0 = t[6]; ts.wrs = t[7]; ts.wcs = t[8];
    return ts;
}

extern "C" int vpd_op_conv2d(const void* x, const void* w, void* y, double* stats, int n, int xHp, int xWp, int xC,
                             int yHp, int yWp, int yC, int ypad, int Hs, int Ws, int osub, int oph, int opw, int istr,
                             int Kc, int Co, const int* tapset9, int accumulate, void* stream) {
    ConvParams q;
    memset(&q, 0, sizeof q);
    q.x = (const bf16_t*)x; q.xHp = xHp; q.xWp = xWp; q.xC = xC; q.w = (const bf16_t*)w;
    q.y = (bf16_t*)y; q.yHp = yHp; q.yWp = yWp; q.yC = yC; q.ypad = ypad; q.stats = stats;
    q.N = n; q.Hs = Hs; q.Ws = Ws; q.osub = osub; q.oph = oph; q.opw = opw; q.istr = istr;
    q.Kc = Kc; q.Co = Co; q.M = n * Hs * Ws; q.accumulate = accumulate;
    q.taps = tapset_from(tapset9);
    if (q.taps.nr < 1 || q.taps.nc < 1) return fail("empty tap set");
    LCHECK(vpd_launch_conv(q, (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_op_conv2d_ep(const void* x, const void* w, void* y, int n, int xHp, int xWp, int xC, int yHp, int yWp,
                                int yC, int ypad, int Hs, int Ws, int istr, int Kc, int Co, const int* tapset9,
                                const float* ep_scale, const float* ep_shift, const void* res_padded, int ep_relu,
                                int accumulate, const unsigned char* acc_mask, void* stream) {
    if ((ep_scale == nullptr) != (ep_shift == nullptr)) return fail("ep_scale and ep_shift come together");
    if (ep_scale && accumulate) return fail("eval epilogue or accumulate, not both");
    if (res_padded && !ep_scale) return fail("a residual needs the eval epilogue");
    if (acc_mask && (!accumulate || ypad != 0 || yC != Co || yHp != Hs || yWp != Ws)) return fail("acc_mask: accumulate onto a dense y");
    ConvParams q;
    memset(&q, 0, sizeof q);
    q.x = (const bf16_t*)x; q.xHp = xHp; q.xWp = xWp; q.xC = xC; q.w = (const bf16_t*)w;
    q.y = (bf16_t*)y; q.yHp = yHp; q.yWp = yWp; q.yC = yC; q.ypad = ypad;
    q.N = n; q.Hs = Hs; q.Ws = Ws; q.osub = 1; q.istr = istr;
    q.Kc = Kc; q.Co = Co; q.M = n * Hs * Ws; q.accumulate = accumulate; q.acc_mask = acc_mask;
    q.ep_scale = ep_scale; q.ep_shift = ep_shift; q.ep_relu = ep_relu;
    if (res_padded) { q.res = (const bf16_t*)res_padded; q.rHp = Hs + 2; q.rWp = Ws + 2; q.rC = Co; q.rpad = 1; }
    q.taps = tapset_from(tapset9);
    if (q.taps.nr < 1 || q.taps.nc < 1) return fail("empty tap set");
    LCHECK(vpd_launch_conv(q, (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_op_conv2d_bnsums(const void* x, const void* w, void* y, const void* bst_z, const unsigned char* bst_mask,
                                    double* rows, int n, int xHp, int xWp, int xC, int Hs, int Ws, int Kc, int Co,
                                    const int* tapset9, int accumulate, void* stream) {
    if (!bst_z || !bst_mask || !rows) return fail("null argument");
    ConvParams q;
    memset(&q, 0, sizeof q);
    q.x = (const bf16_t*)x; q.xHp = xHp; q.xWp = xWp; q.xC = xC; q.w = (const bf16_t*)w;
    q.y = (bf16_t*)y; q.yHp = Hs; q.yWp = Ws; q.yC = Co; q.ypad = 0;
    q.N = n; q.Hs = Hs; q.Ws = Ws; q.osub = 1; q.istr = 1;
    q.Kc = Kc; q.Co = Co; q.M = n * Hs * Ws; q.accumulate = accumulate;
    q.bst_z = (const bf16_t*)bst_z; q.bst_mask = bst_mask; q.stats = rows; q.stat_rows = VPD_FUSED_ROWS;
    q.taps = tapset_from(tapset9);
    if (q.taps.nr < 1 || q.taps.nc < 1) return fail("empty tap set");
    if (!vpd_conv_takes_bn_sums(q)) return fail("no kernel takes the BatchNorm sums for this shape");
    LCHECK(vpd_launch_conv(q, (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_op_bn_forward(const void* z, const double* rows, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float* mean, float* rstd, float* scale, float* shift, const void* res,
                                 void* out, unsigned char* mask_bits, int n, int H, int W, int C, int relu, float momentum,
                                 float eps, void* stream) {
    if (!z || !rows || !gamma || !beta || !mean || !rstd || !scale || !shift || !out) return fail("null argument");
    BnApplyParams a;
    memset(&a, 0, sizeof a);
    a.z = (const bf16_t*)z;
    a.res_kind = res ? 1 : 0; a.res = (const bf16_t*)res; a.rHp = H + 2; a.rWp = W + 2; a.rpad = 1;
    a.out = (bf16_t*)out; a.oHp = H + 2; a.oWp = W + 2; a.opad = 1;
    a.M = n * H * W; a.H = H; a.W = W; a.C = C; a.relu = relu; a.mask_out = mask_bits;
    BnFusedFwd f;
    memset(&f, 0, sizeof f);
    f.rows = const_cast<double*>(rows); f.count = (float)a.M; f.gamma = gamma; f.beta = beta; f.rm = running_mean; f.rv = running_var;
    f.mean = mean; f.rstd = rstd; f.scale = scale; f.shift = shift; f.momentum = momentum; f.eps = eps;
    LCHECK(vpd_launch_bn_fwd_fused(a, f, (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_op_bn_backward_apply(const void* dy, const void* z, const unsigned char* mask_bits, const double* rows,
                                        const float* gamma, const float* mean, const float* rstd, void* dz, float* dgamma,
                                        float* dbeta, int n, int H, int W, int C, void* stream) {
    if (!dy || !z || !mask_bits || !rows || !gamma || !mean || !rstd || !dz || !dgamma || !dbeta) return fail("null argument");
    BnBwdParams b;
    memset(&b, 0, sizeof b);
    b.dy = (const bf16_t*)dy; b.z = (const bf16_t*)z; b.mean = mean; b.rstd = rstd;
    b.dz = (bf16_t*)dz; b.dzHp = H + 2; b.dzWp = W + 2; b.dzpad = 1;
    b.M = n * H * W; b.H = H; b.W = W; b.C = C; b.mask_bits = mask_bits;
    BnFusedBwd f;
    memset(&f, 0, sizeof f);
    f.rows = const_cast<double*>(rows); f.gamma = gamma; f.dgamma = dgamma; f.dbeta = dbeta; f.count = (float)b.M;
    LCHECK(vpd_launch_bn_bwd_apply_fused(b, f, (hipStream_t)stream));
    return 0;
}

extern "C" int vpd_op_wgrad(const void* dz, const void* x, float* dw, int n, int dzHp, int dzWp, int dzC, int dzpad,
                            int xHp, int xWp, int xC, int Hs, int Ws, int istr, int Kc, int Co, const int* tapset9,
                            float* slab, void* stream) {
    WgradParams q;
    memset(&q, 0, sizeof q);
    q.dz = (const bf16_t*)dz; q.dzHp = dzHp; q.dzWp = dzWp; q.dzC = dzC; q.dzpad = dzpad;
    q.x = (const bf16_t*)x; q.xHp = xHp; q.xWp = xWp; q.xC = xC; q.dw = dw; q.slab = slab;
    q.N = n; q.Hs = Hs; q.Ws = Ws; q.istr = istr; q.Kc = Kc; q.Co = Co; q.M = n * Hs * Ws;
    q.taps = tapset_from(tapset9);
    if (q.taps.nr < 1 || q.taps.nc < 1) return fail("empty tap set");
    LCHECK(vpd_launch_wgrad(q, (hipStream_t)stream));
    return 0;
}

// Grouped 128 x 64 weight gradients (conv_wgrad128_persistent_kernel) of `nprob` 3x3 stride-1 pad-1 convolutions in ONE
// launch.  dims: 7 ints per problem {n, H, W, Co, Ci, stride, k} (H, W: OUTPUT size; stride 1 or 2; k = 3: 3x3 pad 1, k = 1: 1x1
// pad 0); dz[i]: padded bf16 [n][H+2][W+2][Co]; x[i]: padded bf16 [n][stride*H+2][stride*W+2][Ci]; dw[i]: fp32 [k*k][Co][Ci]; slab[i]: fp32 scratch of vpd_op_wgrad128_slab_floats(Co, Ci) floats;
// dev_table: vpd_op_wgrad128_table_bytes() of device memory.
extern "C" size_t vpd_op_wgrad128_table_bytes(void) { return vpd_wgrad128_table_bytes(); }
extern "C" size_t vpd_op_wgrad128_slab_floats(int Co, int Ci) { return vpd_wgrad_group_slab_floats(0, Co, Ci, 1) > vpd_wgrad_group_slab_floats(0, Co, Ci, 9) ? vpd_wgrad_group_slab_floats(0, Co, Ci, 1) : vpd_wgrad_group_slab_floats(0, Co, Ci, 9); }
extern "C" int vpd_op_wgrad128_group(int nprob, const void* const* dz, const void* const* x, float* const* dw,
                                     float* const* slab, const int* dims, void* dev_table, void* stream) {
    if (nprob < 1 || nprob > 18 || !dz || !x || !dw || !slab || !dims || !dev_table) return fail("bad argument");
    WgradParams qs[18];
    for (int i = 0; i < nprob; ++i) {
        const int n = dims[7 * i], H = dims[7 * i + 1], W = dims[7 * i + 2], Co = dims[7 * i + 3], Ci = dims[7 * i + 4];
        const int S = dims[7 * i + 5], ksz = dims[7 * i + 6];
        if ((S != 1 && S != 2) || (ksz != 1 && ksz != 3)) return fail("stride must be 1 or 2, k 1 or 3");
        WgradParams q;
        memset(&q, 0, sizeof q);
        q.dz = (const bf16_t*)dz[i]; q.dzHp = H + 2; q.dzWp = W + 2; q.dzC = Co; q.dzpad = 1;
        q.x = (const bf16_t*)x[i]; q.xHp = S * H + 2; q.xWp = S * W + 2; q.xC = Ci;
        q.dw = dw[i]; q.slab = slab[i];
        q.N = n; q.Hs = H; q.Ws = W; q.istr = S; q.Kc = Ci; q.Co = Co; q.M = n * H * W;
        if (ksz == 3) {
            q.taps.nr = 3; q.taps.nc = 3; q.taps.dy0 = 0; q.taps.dys = 1; q.taps.dx0 = 0; q.taps.dxs = 1;
            q.taps.w0 = 0; q.taps.wrs = 3; q.taps.wcs = 1;
        } else {
            q.taps.nr = 1; q.taps.nc = 1; q.taps.dy0 = 1; q.taps.dys = 1; q.taps.dx0 = 1; q.taps.dxs = 1;
            q.taps.w0 = 0; q.taps.wrs = 1; q.taps.wcs = 1;
        }
        if (!vpd_wgrad128_eligible(q)) return fail("shape not eligible for the 128 x 64 weight-gradient kernel");
        qs[i] = q;
    }
    static void* cache = vpd_wgrad128_cache_new();      // one schedule cache for the op entry point (rebuilt when shapes change)
    LCHECK(vpd_launch_wgrad128_group(qs, nprob, cache, dev_table, (hipStream_t)stream));
    return 0;
}
