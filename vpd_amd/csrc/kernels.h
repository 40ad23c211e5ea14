// Host-side launcher declarations for the HIP kernels (internal to libvpdhip).
#pragma once
#include <vector>

#include "common.h"

struct BnApplyParams {
    const bf16_t* z;                    // dense [M][C]
    const float* scale; const float* shift;
    int res_kind;                       // 0 none, 1 padded activation, 2 dense z with rscale/rshift
    const bf16_t* res; int rHp, rWp, rpad;
    const float* rscale; const float* rshift;
    bf16_t* out; int oHp, oWp, opad;    // padded output
    int M, H, W, C, relu;
    unsigned char* mask_out;            // fused forward only: [M][C/8] bytes, bit j of byte (m, c8) = out[m][8*c8 + j] > 0, or null
    int xcd_tile_px;                    // fused forward only: pixel tile of the neighbouring 3x3 launches (vpd_bn_virtual_block) or 0
};

struct StemPoolParams {
    const bf16_t* z; int Hz, Wz;        // dense conv output
    const float* scale; const float* shift;
    bf16_t* out; int opad;              // padded pooled output
    unsigned char* idx;                 // dense argmax (train) or null
    int N, Ho, Wo, C;
};

struct BnBwdParams {
    const bf16_t* dy;                   // dense [M][C] gradient w.r.t. the BN(+ReLU) output
    bf16_t* dy_rw;                      // same buffer, written with g when write_g
    const bf16_t* z;                    // dense conv output saved by forward
    const bf16_t* act; int aHp, aWp, apad;   // padded post-ReLU activation (mask) or null
    const float* mean; const float* rstd;
    float* coef;                        // [3][C] scratch
    double* partials;                   // [VPD_STAT_ROWS][2][C] fp64 accumulator rows
    bf16_t* dz; int dzHp, dzWp, dzpad;  // output (padded or dense)
    int M, H, W, C, write_g, ppb;
    const float* mscale; const float* mshift;   // when act == null and these are set: ReLU mask = (mscale*z + mshift > 0)
    // fused backward only: the ReLU mask as ONE BIT per element ([M][C/8] bytes, written by the fused forward) instead of
    // the stored activation; g is then neither read back from nor written to dy (the consumer of the identity path masks
    // dy itself: ConvParams::acc_mask)
    const unsigned char* mask_bits;
    // fused backward with mask_bits only: dy does not exist yet -- it is the gradient of the global average pool, dy[b][y][x][c] =
    // bf16(dy_pooled[b][c] * dy_pool_scale) (avgpool_bwd_kernel's value).  The kernel computes it, uses it, and writes it to dy_rw
    // (unmasked: the identity path of the block adds onto it later) -- the avgpool_bwd launch in front of it is gone
    const float* dy_pooled; float dy_pool_scale;
    int xcd_tile_px;                    // bn_bwd_apply_fused_kernel only: as BnApplyParams::xcd_tile_px
};

struct StemPoolBwdParams {
    const bf16_t* dpool;                // dense [N][Ho][Wo][C]
    const unsigned char* idx;
    const bf16_t* z;                    // dense [N][Hz][Wz][C]
    const float* mean; const float* rstd; const float* scale; const float* shift;
    bf16_t* g;                          // unused (g is recomputed, never stored)
    double* partials;
    int M, Hz, Wz, Ho, Wo, C, ppb;
    int pass; const float* coef; bf16_t* dz;    // filled by the launcher
    // pooled post-ReLU activation (padded NHWC, border ppad) and the BN affine parameters: the backward sums are taken
    // over the POOLED positions (4x fewer than z pixels), where xhat of the arg-max pixel is (a - beta) / gamma
    const bf16_t* pooled; int ppad; const float* gamma_p; const float* beta_p;
};

struct PackDesc {                       // one convolution's weight tensors
    long long src_off;                  // fp32 OIHW master / grad offset (elements) in the flat buffers
    long long fwd_off;                  // bf16 [ntaps][Co][Kc] offset in the packed-weight arena
    long long dgr_off;                  // bf16 [kh*kw][Ci][Co] offset (dgrad layout) or -1
    long long wg_off;                   // fp32 [ntaps][Co][Kc] offset in the wgrad scratch
    int Co, Ci, kh, kw, Kc, ntaps, stem;   // stem: 1 = 7x7 row-tap packing; 2 = not a conv: plain range (AdamW only)
    long long numel;                    // stem == 2: length of the range at src_off
};
struct AdamHyper { float decay, omb1, b2, omb2, step_size, inv_sqrt_bc2, eps, gscale; };      // gscale: 1 / loss scale (fp16 training), else 1

hipError_t vpd_launch_conv(const ConvParams& p, hipStream_t stream);
hipError_t vpd_launch_scale(float* x, long n, float s, hipStream_t stream);      // head.hip: x *= s (the loss scale on d(loss)/d(pred))
// conv_stream.hip: persistent streaming kernel for 1x1 convolutions with <= 256 input channels on many pixels
bool vpd_conv1x1_stream_eligible(const ConvParams& p);
hipError_t vpd_launch_conv1x1_stream(const ConvParams& p, hipStream_t stream);
int vpd_conv_kernel_class(const ConvParams& p);      // 0..4, see conv_igemm.hip
bool vpd_conv_takes_bn_sums(const ConvParams& p);
// pixels per tile when `p` runs on conv3x3_pws_kernel with its XCD-affine tile order (pixel tile t on XCD t % 8), else 0
int vpd_conv_xcd_tile_px(const ConvParams& p);         // epilogue can take the consuming BatchNorm's backward sums (bst_z)
hipError_t vpd_launch_wgrad_reduce(const WgradParams& p, hipStream_t stream);   // slab sum of a deferred halo wgrad
extern "C" int vpd_conv_bm(int M, int Co);
hipError_t vpd_launch_wgrad(const WgradParams& p, hipStream_t stream);
size_t vpd_wgrad_slab_bytes();
// grouped (per-stage, deferred) weight gradients: see WgGroup in conv_wgrad.hip
bool vpd_wgrad_group_eligible(const WgradParams& p);
size_t vpd_wgrad_group_slab_floats(int M, int Co, int Kc, int ntaps = 9);
hipError_t vpd_launch_wgrad_group(const WgradParams* ps, int n, hipStream_t stream);
bool vpd_wgrad_overwrites(const WgradParams& p);
// 128 x 64 tiles, persistent blocks, host-built schedule (conv_wgrad128_persistent_kernel in conv_wgrad.hip)
bool vpd_wgrad128_eligible(const WgradParams& p);
size_t vpd_wgrad128_table_bytes();
void* vpd_wgrad128_cache_new();
void vpd_wgrad128_cache_free(void* cache);
hipError_t vpd_launch_wgrad128_group(const WgradParams* ps, int n, void* cache, void* dev_table, hipStream_t stream);
bool vpd_wgrad_halo_shape_ok(int Hout, int Wout, int stride = 1, int Hin = 0, int Win = 0);

hipError_t vpd_launch_bn_finalize(double* partials, int T, int C, float count, const float* gamma,
                                  const float* beta, float* rm, float* rv, float momentum, float eps,
                                  float* mean, float* rstd, float* scale, float* shift, hipStream_t s);
hipError_t vpd_launch_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                              float* scale, float* shift, int C, hipStream_t s);
hipError_t vpd_launch_bn_apply(const BnApplyParams& p, hipStream_t s);
hipError_t vpd_launch_stem_pool(const StemPoolParams& p, hipStream_t s);
int vpd_bn_bwd_blocks(int M, int C, int* ppb_out);
hipError_t vpd_launch_bn_bwd(const BnBwdParams& p, float count, const float* gamma, float* dgamma, float* dbeta,
                             hipStream_t s, bool reduce_done = false);
hipError_t vpd_launch_stem_pool_bwd(const StemPoolBwdParams& p, float count, const float* gamma, float* dgamma,
                                    float* dbeta, float* coef, bf16_t* dz, hipStream_t s);

// fused BatchNorm passes (bn.hip): per-BatchNorm accumulator rows [VPD_FUSED_ROWS][2][C] of doubles, zeroed by the caller
struct BnFusedFwd {
    double* rows; float count; const float* gamma; const float* beta; float* rm; float* rv;
    float* mean; float* rstd; float* scale; float* shift;
    double* rows2; float count2; const float* gamma2; const float* beta2; float* rm2; float* rv2;      // residual BN or null
    float* mean2; float* rstd2; float* scale2; float* shift2;
    float momentum, eps;
};
struct BnFusedBwd {
    double* rows; void* sync; unsigned* err;    // sync: VPD_GRID_SYNC_BYTES, zeroed; err: sticky time-out counter
    const float* gamma; float* dgamma; float* dbeta; float count;
};
#define VPD_GRID_SYNC_BYTES (18 * 128)
hipError_t vpd_launch_bn_fwd_fused(const BnApplyParams& p, const BnFusedFwd& f, hipStream_t s);
// conv_stream.hip: a Bottleneck's closing 1x1 convolution with its train-mode BatchNorm, the convolution recomputed instead of
// written and read back (modes: 0 statistics only, 1 forward apply, 2 backward sums, 3 backward apply)
bool vpd_conv1x1_bn_eligible(const ConvParams& p);
// ... of a down-sampling Bottleneck whose closing 1x1 conv (p.x, p.w) and 1x1 branch (p.x2, p.w2) both have 64 input channels, stride 1
bool vpd_conv1x1_bn2_eligible(const ConvParams& p);
hipError_t vpd_launch_conv1x1_bn2(const ConvParams& p, const BnFusedFwd* fwd, const BnFusedBwd* bwd3, const BnFusedBwd* bwdD,
                                  const float* mean3, const float* rstd3, const float* meanD, const float* rstdD,
                                  unsigned char* mask_out, bf16_t* dz3, bf16_t* dzD, int dzpad, int mode, hipStream_t stream);
hipError_t vpd_launch_conv1x1_bn(const ConvParams& p, const BnFusedFwd* fwd, const BnFusedBwd* bwd, const float* mean,
                                 const float* rstd, unsigned char* mask_out, bf16_t* dz, int dzpad, int mode, hipStream_t stream);
bool vpd_bn_bwd_fused_ok(int M, int C, bool mask_act, bool write_g);
hipError_t vpd_launch_bn_bwd_fused(const BnBwdParams& p, const BnFusedBwd& f, hipStream_t s);
// BatchNorm backward whose sums (sum g, sum g * z) the producing data gradient's epilogue has already added to `f.rows`
// (ConvParams::bst_z): finalize + apply in one launch, no reduction pass, no grid barrier.  p.mask_bits is required.
// fB / zB / meanB / rstdB / dzB: a second BatchNorm fed with the same masked gradient (same shape, same padded dz geometry)
hipError_t vpd_launch_bn_bwd_apply_fused(const BnBwdParams& p, const BnFusedBwd& f, hipStream_t s, const BnFusedBwd* fB = nullptr,
                                         const bf16_t* zB = nullptr, const float* meanB = nullptr, const float* rstdB = nullptr,
                                         bf16_t* dzB = nullptr);
// two BatchNorm backwards sharing dy and the ReLU mask (a down-sampling block's conv2 BN + its 1x1 branch's BN) in one launch
bool vpd_bn_bwd_fused2_ok(int M, int C);
hipError_t vpd_launch_bn_bwd_fused2(const BnBwdParams& p, const BnFusedBwd& fA, const BnFusedBwd& fB, const bf16_t* zB,
                                    const float* meanB, const float* rstdB, bf16_t* dzB, hipStream_t s);

// head.hip
hipError_t vpd_launch_avgpool(const bf16_t* act, int Hp, int Wp, int pad, int H, int W, int C, int N, float* pooled,
                              hipStream_t s);
hipError_t vpd_launch_avgpool_bwd(const float* dpooled, int H, int W, int C, int N, bf16_t* dact, hipStream_t s);
// Y[M][N] (=|+=) op(A)[M][K] * op(B)[K][N] (+ bias[N]) (relu?)   fp32, row-major
//   ta: A is stored [K][M];  tb: B is stored [N][K]
hipError_t vpd_launch_sgemm(const float* A, const float* B, float* Y, const float* bias, int M, int N, int K, int ta,
                            int tb, int relu, hipStream_t s);
hipError_t vpd_launch_colsum(const float* A, int M, int N, float* out, hipStream_t s);
hipError_t vpd_launch_relu_mask(float* d, const float* act, long n, hipStream_t s);
hipError_t vpd_launch_mse(const float* e, const float* t, long n, float* de, float* loss_step, double* loss_accum,
                          hipStream_t s);

// optim.hip
hipError_t vpd_launch_pack_input(const float* x, int N, int C, int H, int W, bf16_t* out, int Hp, int Wp, int pad,
                                 int Cp, hipStream_t s);
hipError_t vpd_launch_pack_weights(const PackDesc* d_descs, int ndesc, const int* d_blockmap, int nblocks,
                                   const float* master, bf16_t* arena, hipStream_t s);
hipError_t vpd_launch_adamw_pack(const PackDesc* d_descs, const int* d_blockmap, int nblocks, float* p, const float* g,
                                 float* m, float* v, bf16_t* arena, double lr, double b1, double b2, double eps, double wd,
                                 int step, hipStream_t s, const float* wg = nullptr, float gscale = 1.f);      // wg: conv gradients still in the scratch
hipError_t vpd_launch_unpack_grads(const PackDesc* d_descs, int ndesc, const int* d_blockmap, int nblocks,
                                   const float* wg, float* grads, hipStream_t s);
hipError_t vpd_launch_adamw(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                            double eps, double wd, int step, hipStream_t s, float gscale = 1.f);

#define ZR_MAX 16
struct ZeroRanges {                     // 16-byte aligned ranges, lengths in float4
    float* ptr[ZR_MAX];
    long n4[ZR_MAX];
    int count;
};
hipError_t vpd_launch_zero_ranges(const ZeroRanges& z, hipStream_t s);

// augment.hip (declared with the public vpd_aug_params of include/vpd_hip.h)
struct vpd_aug_params;
hipError_t vpd_launch_augment(const unsigned char* rgb, const unsigned char* flow, const unsigned char* mask,
                              const float* noise, const vpd_aug_params* params, int N, int H, int W, int out_dim,
                              const float* mean_std6, float noise_sd, float* out_nchw, bf16_t* xin, int xHp, int xWp,
                              int xpad, float* cmean_scratch, hipStream_t s);
hipError_t vpd_launch_views(const unsigned char* rgb, const unsigned char* flow, int F, int K, int H, int W,
                            const float* mean_std6, bf16_t* xin, int xHp, int xWp, int xpad, hipStream_t s);
