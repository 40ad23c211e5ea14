// Train-time input pipeline on the device: decoded u8 crops -> the network input, in one pass.
// Replaces, per item, reference vpd_dataset/common.py:52-69 (u8 -> float, ColorJitter, Normalize, flow decode),
// vpd_dataset/single_frame.py:178-203 (mask noise, concat, h-flip with x-flow negation) and common.py:49-50/:80
// (RandomResizedCrop = crop + bilinear resize).  The random decisions are made on the host (vpd_aug_params, one
// per crop); the kernels are deterministic functions of (pixels, params).
//
// Two launches: (1) per-crop mean of the grey image as it is when ColorJitter's contrast op runs (that op blends
// with a whole-image mean, the only non-local step); (2) one thread per OUTPUT pixel: its four bilinear source
// pixels are fetched as u8, pushed through jitter / normalise / noise / flip, and blended.  Output goes to an fp32
// NCHW batch (the reference's batch['img'] contract) and/or straight into the stem's bf16 padded NHWC staging
// buffer (16-byte store per pixel; the fp32 batch is then never materialised).
//
// HBM-bound by construction: 5 B read per source pixel, 16 B (bf16 x 8) or 20 B (fp32 x 5) written per output pixel.
#include "../../include/vpd_hip.h"
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)      // keep torchvision's operation order (no fused multiply-add)

namespace {

struct Rgb { float r, g, b; };

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float blend(float a, float b, float ratio) { return clamp01(ratio * a + (1.0f - ratio) * b); }
__device__ __forceinline__ float grey(const Rgb& c) { return 0.2989f * c.r + 0.587f * c.g + 0.114f * c.b; }

// torchvision adjust_hue on one pixel: _rgb2hsv, h = (h + f) mod 1, _hsv2rgb
__device__ __forceinline__ Rgb hue_shift(const Rgb& c, float f) {
    const float maxc = fmaxf(c.r, fmaxf(c.g, c.b));
    const float minc = fminf(c.r, fminf(c.g, c.b));
    const bool eqc = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eqc ? 1.f : maxc);
    const float div = eqc ? 1.f : cr;
    const float rc = (maxc - c.r) / div, gc = (maxc - c.g) / div, bc = (maxc - c.b) / div;
    const float hr = (maxc == c.r) ? (bc - gc) : 0.f;
    const float hg = ((maxc == c.g) && (maxc != c.r)) ? (2.0f + rc - bc) : 0.f;
    const float hb = ((maxc != c.g) && (maxc != c.r)) ? (4.0f + gc - rc) : 0.f;
    float h = hr + hg + hb;
    h = fmodf(h / 6.0f + 1.0f, 1.0f);
    h = h + f;
    h = h - floorf(h);                                  // python's % 1.0
    const float v = maxc;
    const float h6 = h * 6.0f;
    const float fl = floorf(h6);
    const float fr = h6 - fl;
    int i = (int)fl;
    const float p = clamp01(v * (1.0f - s));
    const float q = clamp01(v * (1.0f - s * fr));
    const float t = clamp01(v * (1.0f - (s * (1.0f - fr))));
    i = i % 6;
    Rgb o;
    switch (i) {
        case 0: o.r = v; o.g = t; o.b = p; break;
        case 1: o.r = q; o.g = v; o.b = p; break;
        case 2: o.r = p; o.g = v; o.b = t; break;
        case 3: o.r = p; o.g = q; o.b = v; break;
        case 4: o.r = t; o.g = p; o.b = v; break;
        default: o.r = v; o.g = p; o.b = q; break;
    }
    return o;
}

// ColorJitter ops order[first .. last) on one pixel; `cmean` = grey mean for the contrast op
__device__ __forceinline__ Rgb jitter(Rgb c, const vpd_aug_params& a, int first, int last, float cmean) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int op = (k >= first && k < last) ? a.order[k] : -1;
        if (op == 0) {
            c.r = blend(c.r, 0.f, a.factor[0]); c.g = blend(c.g, 0.f, a.factor[0]); c.b = blend(c.b, 0.f, a.factor[0]);
        } else if (op == 1) {
            c.r = blend(c.r, cmean, a.factor[1]); c.g = blend(c.g, cmean, a.factor[1]); c.b = blend(c.b, cmean, a.factor[1]);
        } else if (op == 2) {
            const float gy = grey(c);
            c.r = blend(c.r, gy, a.factor[2]); c.g = blend(c.g, gy, a.factor[2]); c.b = blend(c.b, gy, a.factor[2]);
        } else if (op == 3) {
            c = hue_shift(c, a.factor[3]);
        }
    }
    return c;
}

__device__ __forceinline__ Rgb load_rgb(const unsigned char* p) {
    Rgb c;
    c.r = (float)p[0] / 255.f; c.g = (float)p[1] / 255.f; c.b = (float)p[2] / 255.f;
    return c;
}

// (1) grey mean of every crop at the point where its contrast op runs.  AUG_MEAN_PARTS blocks per crop write partial
// sums; the consumer adds them in a fixed order (deterministic, unlike atomics).
#define AUG_MEAN_PARTS 8
__global__ __launch_bounds__(256) void aug_contrast_mean_kernel(const unsigned char* rgb, const vpd_aug_params* params,
                                                                int HW, float* partial) {
    __shared__ float sh[4];
    const int n = blockIdx.y, part = blockIdx.x;
    const vpd_aug_params a = params[n];
    int kc = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (a.order[k] == 1) kc = k;
    if (kc < 0) {
        if (threadIdx.x == 0) partial[n * AUG_MEAN_PARTS + part] = 0.f;
        return;
    }
    const unsigned char* img = rgb + (size_t)n * HW * 3;
    const int per = (HW + AUG_MEAN_PARTS - 1) / AUG_MEAN_PARTS;
    const int beg = part * per;
    const int end = beg + per < HW ? beg + per : HW;
    float acc = 0.f;
    for (int i = beg + threadIdx.x; i < end; i += 256) acc += grey(jitter(load_rgb(img + 3 * (size_t)i), a, 0, kc, 0.f));
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[n * AUG_MEAN_PARTS + part] = sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float aug_contrast_mean(const float* partial, int n, int HW) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < AUG_MEAN_PARTS; ++i) t += partial[n * AUG_MEAN_PARTS + i];
    return t / (float)HW;
}

// Philox4x32-10 (Salmon et al. 2011): counter-based, so a source pixel's noise does not depend on who asks for it
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                           unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void normal3(unsigned seed_lo, unsigned seed_hi, unsigned n, unsigned pix, float z[3]) {
    unsigned u[4];
    philox4x32(pix, n, 0u, 0u, seed_lo, seed_hi, u);
    const float a0 = ((float)(u[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), a1 = (float)(u[1] >> 8) * (1.0f / 16777216.0f);
    const float b0 = ((float)(u[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), b1 = (float)(u[3] >> 8) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(a0)), r1 = sqrtf(-2.0f * logf(b0));
    z[0] = r0 * cosf(6.28318530717958648f * a1);
    z[1] = r0 * sinf(6.28318530717958648f * a1);
    z[2] = r1 * cosf(6.28318530717958648f * b1);
}

struct AugArgs {
    const unsigned char* rgb; const unsigned char* flow; const unsigned char* mask; const float* noise;
    const vpd_aug_params* params; const float* cmean;
    int N, H, W, out_dim, C;                 // C = 3 or 5 output channels
    float mean[3], std[3], noise_sd;
    float* out_nchw;                         // [N][C][out][out] or null
    bf16_t* xin; int xHp, xWp, xpad;         // [N][xHp][xWp][8] bf16 (border xpad) or null
};

// the 5 values of source pixel (y, x) of the flipped image of crop n, after jitter / normalise / noise / flow decode
__device__ __forceinline__ void source_pixel(const AugArgs& g, const vpd_aug_params& a, int n, float cmean, int y, int x,
                                             float v[5]) {
    const int xo = a.flip ? g.W - 1 - x : x;
    const size_t pix = ((size_t)n * g.H + y) * g.W + xo;
    Rgb c = jitter(load_rgb(g.rgb + pix * 3), a, 0, 4, cmean);
    v[0] = (c.r - g.mean[0]) / g.std[0];
    v[1] = (c.g - g.mean[1]) / g.std[1];
    v[2] = (c.b - g.mean[2]) / g.std[2];
    if (a.noise && g.mask && g.mask[pix] != 0) {
        float z[3];
        if (g.noise) {
            const size_t hw = (size_t)g.H * g.W, o = (size_t)y * g.W + xo;
            z[0] = g.noise[((size_t)n * 3 + 0) * hw + o];
            z[1] = g.noise[((size_t)n * 3 + 1) * hw + o];
            z[2] = g.noise[((size_t)n * 3 + 2) * hw + o];
        } else {
            normal3(a.seed_lo, a.seed_hi, (unsigned)n, (unsigned)(y * g.W + xo), z);
        }
        v[0] = v[0] + z[0] * g.noise_sd; v[1] = v[1] + z[1] * g.noise_sd; v[2] = v[2] + z[2] * g.noise_sd;
    }
    if (g.C == 5) {
        // (u8 / 255) - 0.5 in double, then rounded to float: what numpy + FloatTensor do in the reference
        float fx = (float)((double)g.flow[pix * 2 + 0] / 255.0 - 0.5);
        const float fy = (float)((double)g.flow[pix * 2 + 1] / 255.0 - 0.5);
        if (a.flip) fx = -fx;
        v[3] = fx; v[4] = fy;
    } else {
        v[3] = 0.f; v[4] = 0.f;
    }
}

// torch upsample_bilinear2d, align_corners = false: source index and weights of output index d
__device__ __forceinline__ void bilinear_axis(int d, int in_size, int out_size, int* i0, int* i1, float* l0, float* l1) {
    const float scale = (float)in_size / (float)out_size;
    float src = scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int a = (int)src;
    *i0 = a;
    *i1 = a + 1 < in_size ? a + 1 : in_size - 1;
    float w1 = src - (float)a;
    w1 = fminf(fmaxf(w1, 0.f), 1.f);
    *l1 = w1;
    *l0 = 1.f - w1;
}

#define AUG_ROWS 4            // output rows per block
#define AUG_SRC_ROWS 8        // staged source rows (enough for any up-sampling and down-sampling up to 2x)
#define AUG_SRC_W 256         // widest staged crop window

__device__ __forceinline__ void aug_store(const AugArgs& g, int n, int oy, int ox, const float o[5]) {
    if (g.out_nchw) {
        const size_t plane = (size_t)g.out_dim * g.out_dim;
        float* dst = g.out_nchw + (size_t)n * g.C * plane + (size_t)oy * g.out_dim + ox;
        for (int c = 0; c < g.C; ++c) dst[c * plane] = o[c];
    }
    if (g.xin) {
        float v8[8] = {o[0], o[1], o[2], o[3], o[4], 0.f, 0.f, 0.f};
        *reinterpret_cast<uint4*>(g.xin + (((size_t)n * g.xHp + oy + g.xpad) * g.xWp + ox + g.xpad) * 8) = pack8(v8);
    }
}

// One block = AUG_ROWS output rows of one crop.  The source rows they interpolate from are transformed ONCE into
// LDS (jitter + hue cost ~1.5 k instructions per source pixel: the kernel is ALU-bound, so each source pixel must
// not be recomputed by its four consumers), then every thread blends from LDS.
__global__ __launch_bounds__(128) void aug_apply_kernel(const AugArgs g) {
    __shared__ float src[AUG_SRC_ROWS][AUG_SRC_W][5];
    const int n = blockIdx.y;
    const int oy0 = blockIdx.x * AUG_ROWS;
    const vpd_aug_params a = g.params[n];
    const float cmean = aug_contrast_mean(g.cmean, n, g.H * g.W);
    int oy_last = oy0 + AUG_ROWS - 1;
    oy_last = oy_last < g.out_dim ? oy_last : g.out_dim - 1;
    if (a.crop_h == g.out_dim && a.crop_w == g.out_dim) {          // no resize: exact copy of the window
        for (int oy = oy0; oy <= oy_last; ++oy)
            for (int ox = threadIdx.x; ox < g.out_dim; ox += 128) {
                float o[5];
                source_pixel(g, a, n, cmean, a.crop_i + oy, a.crop_j + ox, o);
                aug_store(g, n, oy, ox, o);
            }
        return;
    }
    int ya, yb, t0, t1;
    float l0, l1;
    bilinear_axis(oy0, a.crop_h, g.out_dim, &ya, &t1, &l0, &l1);
    bilinear_axis(oy_last, a.crop_h, g.out_dim, &t0, &yb, &l0, &l1);
    const int nrows = yb - ya + 1;
    const bool staged = nrows <= AUG_SRC_ROWS && a.crop_w <= AUG_SRC_W;      // block-uniform
    if (staged) {
        for (int i = threadIdx.x; i < nrows * a.crop_w; i += 128) {
            const int r = i / a.crop_w, x = i - r * a.crop_w;
            float v[5];
            source_pixel(g, a, n, cmean, a.crop_i + ya + r, a.crop_j + x, v);
#pragma unroll
            for (int c = 0; c < 5; ++c) src[r][x][c] = v[c];
        }
        __syncthreads();
    }
    for (int oy = oy0; oy <= oy_last; ++oy) {
        int y0, y1;
        float ly0, ly1;
        bilinear_axis(oy, a.crop_h, g.out_dim, &y0, &y1, &ly0, &ly1);
        for (int ox = threadIdx.x; ox < g.out_dim; ox += 128) {
            int x0, x1;
            float lx0, lx1;
            bilinear_axis(ox, a.crop_w, g.out_dim, &x0, &x1, &lx0, &lx1);
            float v00[5], v01[5], v10[5], v11[5], o[5];
            if (staged) {
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    v00[c] = src[y0 - ya][x0][c]; v01[c] = src[y0 - ya][x1][c];
                    v10[c] = src[y1 - ya][x0][c]; v11[c] = src[y1 - ya][x1][c];
                }
            } else {
                source_pixel(g, a, n, cmean, a.crop_i + y0, a.crop_j + x0, v00);
                source_pixel(g, a, n, cmean, a.crop_i + y0, a.crop_j + x1, v01);
                source_pixel(g, a, n, cmean, a.crop_i + y1, a.crop_j + x0, v10);
                source_pixel(g, a, n, cmean, a.crop_i + y1, a.crop_j + x1, v11);
            }
#pragma unroll
            for (int c = 0; c < 5; ++c)
                o[c] = ly0 * (lx0 * v00[c] + lx1 * v01[c]) + ly1 * (lx0 * v10[c] + lx1 * v11[c]);
            aug_store(g, n, oy, ox, o);
        }
    }
}

}  // namespace

hipError_t vpd_launch_augment(const unsigned char* rgb, const unsigned char* flow, const unsigned char* mask,
                              const float* noise, const vpd_aug_params* params, int N, int H, int W, int out_dim,
                              const float* mean_std6, float noise_sd, float* out_nchw, bf16_t* xin, int xHp, int xWp,
                              int xpad, float* cmean_scratch, hipStream_t s) {
    hipLaunchKernelGGL(aug_contrast_mean_kernel, dim3(AUG_MEAN_PARTS, N), dim3(256), 0, s, rgb, params, H * W, cmean_scratch);
    AugArgs g;
    g.rgb = rgb; g.flow = flow; g.mask = mask; g.noise = noise; g.params = params; g.cmean = cmean_scratch;
    g.N = N; g.H = H; g.W = W; g.out_dim = out_dim; g.C = flow ? 5 : 3;
    for (int i = 0; i < 3; ++i) { g.mean[i] = mean_std6[i]; g.std[i] = mean_std6[3 + i]; }
    g.noise_sd = noise_sd;
    g.out_nchw = out_nchw; g.xin = xin; g.xHp = xHp; g.xWp = xWp; g.xpad = xpad;
    hipLaunchKernelGGL(aug_apply_kernel, dim3((out_dim + AUG_ROWS - 1) / AUG_ROWS, N), dim3(128), 0, s, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Inference views of decoded u8 frames (vpd_dataset/single_frame.py:377-400: [frame, h-flipped frame], no resize, no jitter):
// the generic kernel above spends ~200 instructions per pixel on byte loads and fp32 / fp64 divisions (244 us per 1,000 views
// of 128 x 128: 1.4 TB/s).  A u8 channel takes 256 values, so the normalisation is a table: each block evaluates the SAME
// expressions as source_pixel() once per possible byte ((u / 255 - mean) / std in fp32; u / 255 - 0.5 in double, rounded to
// float), rounds to bf16 as aug_store() does, and the pixels are five LDS look-ups -- bit-identical output, memory-bound.
// One thread produces four consecutive output pixels of one view (64 bytes of the staging row).
// ---------------------------------------------------------------------------
struct ViewArgs {
    const unsigned char* rgb; const unsigned char* flow;      // [F][H][W][3], [F][H][W][2] or null
    int F, K, H, W;                                           // K views per frame: view k = 1 is the h-flip
    float mean[3], std[3];
    bf16_t* xin; int xHp, xWp, xpad;                          // [F*K][xHp][xWp][8]
};
__global__ __launch_bounds__(256) void aug_views_kernel(const ViewArgs g) {
    __shared__ unsigned short lut[5][256];
    {
        const int t = threadIdx.x;                            // blockDim.x == 256
        const float c = (float)t / 255.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) lut[k][t] = f2bf((c - g.mean[k]) / g.std[k]);
        const float f = (float)((double)t / 255.0 - 0.5);
        lut[3][t] = f2bf(f);                                  // x-flow (the flipped view negates it: sign bit)
        lut[4][t] = lut[3][t];
    }
    __syncthreads();
    const int view = blockIdx.y;
    const int frame = view / g.K;
    const bool flip = (view - frame * g.K) == 1;
    const int q = blockIdx.x * 256 + threadIdx.x;             // quad of output pixels inside the view
    const int wq = g.W >> 2;
    if (q >= g.H * wq) return;
    const int y = q / wq, x0 = (q - y * wq) << 2;
    const size_t row = ((size_t)frame * g.H + y) * g.W;
    bf16_t* dst = g.xin + (((size_t)view * g.xHp + y + g.xpad) * g.xWp + x0 + g.xpad) * 8;
    // the four source pixels are one aligned 12-byte (RGB) and one aligned 8-byte (flow) block -- of the mirrored position in
    // the flipped view, read back to front: five dword loads per thread instead of twenty byte loads
    const int xb = flip ? g.W - 4 - x0 : x0;
    const unsigned* pr = reinterpret_cast<const unsigned*>(g.rgb + (row + xb) * 3);
    const unsigned r0 = pr[0], r1 = pr[1], r2 = pr[2];
    unsigned f0 = 0, f1 = 0;
    if (g.flow) {
        const unsigned* pf = reinterpret_cast<const unsigned*>(g.flow + (row + xb) * 2);
        f0 = pf[0]; f1 = pf[1];
    }
    const unsigned long long rlo = (unsigned long long)r0 | ((unsigned long long)r1 << 32);      // bytes 0..7 of the RGB block
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = flip ? 3 - i : i;                       // pixel of the block that lands at output position x0 + i
        const unsigned cr = j * 3 < 8 ? (unsigned)(rlo >> (j * 24)) & 0xffu : (r2 >> ((j * 3 - 8) * 8)) & 0xffu;
        unsigned cg, cb;
        {
            const int bg = j * 3 + 1, bb = j * 3 + 2;
            cg = bg < 8 ? (unsigned)(rlo >> (bg * 8)) & 0xffu : (r2 >> ((bg - 8) * 8)) & 0xffu;
            cb = bb < 8 ? (unsigned)(rlo >> (bb * 8)) & 0xffu : (r2 >> ((bb - 8) * 8)) & 0xffu;
        }
        uint4 o;
        o.x = (unsigned)lut[0][cr] | ((unsigned)lut[1][cg] << 16);
        unsigned fx = 0, fy = 0;
        if (g.flow) {
            const unsigned fw = j < 2 ? f0 : f1;
            const unsigned ux = (fw >> ((j & 1) * 16)) & 0xffu, uy = (fw >> ((j & 1) * 16 + 8)) & 0xffu;
            fx = lut[3][ux];
            if (flip) fx ^= 0x8000u;
            fy = lut[4][uy];
        }
        o.y = (unsigned)lut[2][cb] | (fx << 16);
        o.z = fy;
        o.w = 0u;
        *reinterpret_cast<uint4*>(dst + i * 8) = o;
    }
}
hipError_t vpd_launch_views(const unsigned char* rgb, const unsigned char* flow, int F, int K, int H, int W,
                            const float* mean_std6, bf16_t* xin, int xHp, int xWp, int xpad, hipStream_t s) {
    ViewArgs g;
    g.rgb = rgb; g.flow = flow; g.F = F; g.K = K; g.H = H; g.W = W;
    for (int i = 0; i < 3; ++i) { g.mean[i] = mean_std6[i]; g.std[i] = mean_std6[3 + i]; }
    g.xin = xin; g.xHp = xHp; g.xWp = xWp; g.xpad = xpad;
    const int quads = H * (W / 4);
    hipLaunchKernelGGL(aug_views_kernel, dim3((quads + 255) / 256, F * K), dim3(256), 0, s, g);
    return hipGetLastError();
}
