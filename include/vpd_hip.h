/* libvpdhip -- C ABI of the MI355X-native VPD student train / apply path.
 *
 * The reference (jhong93/vpd) has no FFI: its boundary for this path is a Python
 * object surface.  Each entry point below names the reference
 * code whose device work it replaces; vpd_amd/ (Python) mirrors the reference
 * classes on top of these calls and INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer owned by the caller (torch tensors'
 *     data_ptr()); the library never allocates or frees caller-visible memory;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call only enqueues work on it (no host sync), so calls are hipGraph-capturable;
 *   - return value 0 = ok, negative = error; vpd_last_error() gives the message
 *     (thread-local).  Shape/channel violations are the wrapper's AssertionError
 *     (models/rgb.py:80-82), not error codes.
 *   - flat buffers: `params`/`grads`/`adam_m`/`adam_v` are fp32 arrays of
 *     vpd_plan_param_numel() elements holding every trainable tensor in the
 *     reference's state_dict order and native layout (conv OIHW, linear [out][in]):
 *     encoder tensors first, then the motion decoder's.  `bn_running` holds
 *     running_mean / running_var of every BatchNorm in module order.
 */
#ifndef VPD_HIP_H
#define VPD_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vpd_plan vpd_plan_t;

const char* vpd_last_error(void);
int vpd_abi_version(void);
/* Element type of activations / packed weights / activation gradients this library was built with: "bf16" (libvpdhip.so: training and
 * inference) or "fp16" (libvpdhip_f16.so, the same sources with -DVPD_ELEM_F16).
 * fp16 is the reference's own GPU precision (torch.cuda.amp.autocast + GradScaler: train_vpd_model.py:79,105; models/util.py:55-57);
 * fp16 TRAINING uses a loss scale (vpd_plan_set_loss_scale). */
const char* vpd_elem_dtype(void);

/* Network + workspace description for one (arch, input, head) configuration.
 * Replaces the module construction of RGBF_EmbeddingModel.__init__ (models/rgb.py:46-66),
 * ResNet.__init__/_make_layer (models/module.py:35-110) and, when motion != 0,
 * FCNet(emb_dim,[128,128],2*emb_dim) (train_vpd_model.py:61-65).
 * arch: "resnet18" | "resnet34" (BasicBlock) | "resnet50" | "resnet101" | "wide_resnet50_2" | "wide_resnet101_2"
 * (Bottleneck) -- the ResNet entries of ENCODER_ARCH (models/module.py:17-32).  train != 0 reserves the
 * activations / gradients a train step needs. */
int vpd_plan_create(const char* arch, int c_in, int img_h, int img_w, int emb_dim, int motion,
                    int max_batch, int train, vpd_plan_t** out);
/* `train`: 0 = inference plan, 1 = train plan; OR-ed with VPD_TRAIN_EARLY_BUCKET0 for data-parallel runs: the weight gradients of
 * layer4 are launched at the end of layer4's backward instead of together with layer3's, so gradient bucket 0 (fc + layer4 +
 * motion head, 61 % of the bytes) is final -- and its event recorded -- with three quarters of backward still ahead. */
#define VPD_TRAIN_EARLY_BUCKET0 2
void vpd_plan_destroy(vpd_plan_t* plan);

/* Trainable-tensor table in reference state_dict order (SURVEY.md 8b "state_dict schema").
 * kind: 0 conv weight OIHW, 1 BN weight, 2 BN bias, 3 linear weight [out][in], 4 linear bias.
 * is_decoder: 1 for the motion head's tensors (state_dict keys layers.{0,2,5}.*). */
int vpd_plan_num_tensors(const vpd_plan_t* plan);
int vpd_plan_tensor_info(const vpd_plan_t* plan, int i, int* kind, int* is_decoder, long long* offset,
                         long long* numel, int* ndim, int dims[4]);
long long vpd_plan_param_numel(const vpd_plan_t* plan);        /* multiple of 4 (tail padding) */
/* BatchNorm running statistics: BN i has C channels, running_mean at rm_off, running_var at rv_off. */
int vpd_plan_num_bn(const vpd_plan_t* plan);
int vpd_plan_bn_info(const vpd_plan_t* plan, int i, int* channels, long long* rm_off, long long* rv_off);
long long vpd_plan_bn_numel(const vpd_plan_t* plan);

/* Gradient buckets for data-parallel all-reduce (no reference counterpart: the
 * reference is single-device).  Bucket 0 is complete first during backward. */
int vpd_plan_num_buckets(const vpd_plan_t* plan);
int vpd_plan_bucket_range(const vpd_plan_t* plan, int bucket, long long* offset, long long* numel);

size_t vpd_plan_workspace_bytes(const vpd_plan_t* plan);
/* Zero the workspace (activation borders rely on it) and upload descriptor tables. */
int vpd_plan_init_workspace(vpd_plan_t* plan, void* workspace, void* stream);

/* fp32 master weights -> packed bf16 tap-major weights (+ dgrad layout), and the
 * eval-mode BN fold (scale/shift from running stats).  Call after every change
 * of `params` / `bn_running` (optimizer step, load_state_dict).  Replaces the
 * implicit weight reads of every conv/BN module call. */
int vpd_pack_weights(vpd_plan_t* plan, const float* params, const float* bn_running, void* workspace, void* stream);

/* Eval-mode forward: RGBF_EmbeddingModel.forward under eval()/no_grad, i.e. the
 * device part of embed() (models/rgb.py:72-86) and of ModelTrainer.epoch with
 * optimizer=None (train_vpd_model.py:70-76).  x: f32 [N][c_in][H][W] (NCHW, the
 * DataLoader batch layout, vpd_dataset/single_frame.py:206); emb_out: f32 [N][emb_dim].
 * If target != NULL also runs the motion head (if any) and adds sum-MSE to the
 * loss outputs (train_vpd_model.py:85-87, :93). */
int vpd_forward_eval(vpd_plan_t* plan, const float* params, const float* x, int n, float* emb_out,
                     const float* target, float* loss_step, double* loss_accum, void* workspace, void* stream);

/* x == NULL in vpd_forward_eval / vpd_forward_train: the input batch was already written to the plan's stem staging
 * buffer by vpd_plan_stage_crops (below); the fp32 NCHW batch is then never materialised. */

/* Train-mode forward + loss: encoder(img) -> [fcn_time] -> F.mse_loss(reduction='sum')
 * (train_vpd_model.py:83-88) with train-mode BatchNorm (batch statistics, running-stat
 * update momentum 0.1).  target: f32 [N][emb_dim or 2*emb_dim].  loss_step[0] = this
 * batch's sum-MSE; loss_accum[0] += it (the epoch accumulator of :93 kept on device).
 * n == 0 is legal (a data-parallel rank whose shard of a ragged last batch is empty): loss_step[0] = 0, nothing else
 * is touched; the matching vpd_backward(n = 0) zeroes `grads` and records every bucket event, so the rank still joins
 * the gradient all-reduce with zeros. */
int vpd_forward_train(vpd_plan_t* plan, const float* params, float* bn_running, const float* x,
                      const float* target, int n, float* emb_out, float* loss_step, double* loss_accum,
                      void* workspace, void* stream);

/* loss.backward() of models/util.py:52 for the graph built by vpd_forward_train:
 * writes d loss / d param for every trainable tensor into `grads` (overwrites;
 * the reference zero_grad()s after every step, models/util.py:58).
 * bucket_events: optional array of vpd_plan_num_buckets() hipEvent_t handles, each
 * recorded on `stream` as soon as that bucket's range of `grads` is final. */
int vpd_backward(vpd_plan_t* plan, const float* params, float* grads, int n, void** bucket_events,
                 void* workspace, void* stream);

/* optimizer.step() of models/util.py:53 for torch.optim.AdamW(lr) with torch defaults
 * (train_vpd_model.py:104): decoupled weight decay on every tensor. `step` is 1-based. */
int vpd_adamw_step(float* params, const float* grads, float* adam_m, float* adam_v, long long numel,
                   double lr, double beta1, double beta2, double eps, double weight_decay, int step, void* stream);

/* The same optimizer.step(), fused with the refresh of `plan`'s packed bf16 weights (what vpd_pack_weights would do on
 * the next forward): one pass over params / grads / adam_m / adam_v (16-byte aligned, numel >= the plan's
 * vpd_plan_param_numel; tensors beyond the plan's are updated too).  After it `plan` needs no vpd_pack_weights until
 * the parameters are written by someone else; other plans over the same parameters (the eval plan) still do. */
int vpd_plan_adamw_step(vpd_plan_t* plan, float* params, const float* grads, float* adam_m, float* adam_v,
                        long long numel, double lr, double beta1, double beta2, double eps, double weight_decay,
                        int step, void* workspace, void* stream);

/* Loss scale of this plan's following backward passes and optimizer steps (default 1 = none).  Replaces torch.cuda.amp.GradScaler
 * (train_vpd_model.py:105; scaler.scale(loss).backward() / scaler.step(optimizer): models/util.py:55-57) for fp16 training on
 * libvpdhip_f16.so: vpd_backward multiplies d(loss)/d(pred) by `scale` -- every gradient it produces is scale x its value --
 * and vpd_plan_adamw_step reads gradients x 1 / scale.  A caller that reads the flat gradient buffer itself divides by the scale. */
int vpd_plan_set_loss_scale(vpd_plan_t* plan, float scale);

/* Lazy gradients for the fused train step (reference: models/util.py:50-58, where nothing looks at .grad between
 * loss.backward() and optimizer.step()).  vpd_plan_set_lazy_grads(plan, 1) arms the NEXT vpd_backward: the conv weight
 * gradients then stay in the kernels' own fp32 scratch layout and vpd_plan_adamw_step reads them there, so the layout pass
 * into `grads` (170 MB of traffic per step) is skipped; every other tensor's gradient, and the stem's, is in `grads` as usual.
 * vpd_plan_grads_pending() tells whether `grads` is incomplete; vpd_plan_materialize_grads() completes it on demand (no-op
 * otherwise).  Under data parallelism (bucket events given) a lazy backward leaves bucket b's conv gradients in the workspace
 * range vpd_plan_bucket_scratch_range(plan, b) (byte offset into the workspace, fp32 count): a SUM all-reduce is layout-
 * agnostic (train_vpd_model.py:87: the loss is a sum over crops), so the reducer sums that range and the non-conv tensors
 * (+ the stem) of the flat buffer; bucket b's event is recorded when both are final.  While vpd_plan_grads_pending() == 1 an
 * all-reduce of the FLAT buffer's bucket ranges is invalid: the conv ranges of `grads` are stale and vpd_plan_adamw_step reads
 * the scratch -- a reducer must pick its buffers by vpd_plan_grads_pending(), not by its caller's word (vpd_amd/ddp.py does). */
int vpd_plan_set_lazy_grads(vpd_plan_t* plan, int on);
int vpd_plan_bucket_scratch_range(const vpd_plan_t* plan, int bucket, long long* ws_byte_offset, long long* numel);
int vpd_plan_grads_pending(const vpd_plan_t* plan);
int vpd_plan_materialize_grads(vpd_plan_t* plan, float* grads, void* workspace, void* stream);

/* hipGraph-captured eval forward for a fixed batch size (apply_vpd_model.py:152-168 inner
 * loop at BATCH_SIZE crops per call).  Capture binds the pointers given here. */
/* ---- train-time input pipeline on the device (the step right before the hot path) ----
 * One item of GenericDataset.__getitem__ (vpd_dataset/single_frame.py:168-206) from the decoded PNGs on:
 * u8 RGB -> /255 -> ColorJitter(brightness .2, contrast .2, saturation .05, hue .05) -> Normalize(mean, std)
 * (vpd_dataset/common.py:52-60, :87-92), mask noise (single_frame.py:178-191), flow decode u8/255 - 0.5
 * (common.py:62-69), concat + h-flip with x-flow negation (single_frame.py:193-203), RandomResizedCrop = crop +
 * bilinear resize to out_dim (common.py:49-50, :80).  The random DECISIONS are the caller's (one vpd_aug_params per
 * crop, sampled on the host in torchvision's order of draws); the device work is deterministic given them. */
typedef struct vpd_aug_params {
    int order[4];        /* ColorJitter ops in application order: 0 brightness, 1 contrast, 2 saturation, 3 hue; -1 = none */
    float factor[4];     /* brightness, contrast, saturation factors, hue shift (indexed by op id) */
    int flip;            /* h-flip (and negate flow x) */
    int noise;           /* add noise_sd * N(0,1) to the normalised RGB where mask != 0 */
    int crop_i, crop_j, crop_h, crop_w;   /* crop window (top, left, height, width) in the flipped image */
    unsigned int seed_lo, seed_hi;        /* Philox key of the device noise generator (when noise == NULL) */
} vpd_aug_params;

/* rgb_u8 [n][height][width][3] (RGB), flow_u8 [n][height][width][2] (x, y) or NULL (3-channel model), mask_u8
 * [n][height][width] or NULL, noise f32 [n][3][height][width] standard-normal draws or NULL (device Philox),
 * params: DEVICE array of n; mean_std6: HOST array {mean r,g,b, std r,g,b}; scratch: 8 * n floats on the device.
 * out_nchw: f32 [n][3 or 5][out_dim][out_dim] = the reference's batch['img'] (single_frame.py:206). */
int vpd_augment_crops(const unsigned char* rgb_u8, const unsigned char* flow_u8, const unsigned char* mask_u8,
                      const float* noise, const vpd_aug_params* params, int n, int height, int width, int out_dim,
                      const float* mean_std6, float noise_sd, float* out_nchw, float* scratch, void* stream);
/* Same pipeline, written straight into the plan's stem staging buffer (bf16 NHWC, zero border): follow with
 * vpd_forward_train / vpd_forward_eval with x == NULL.  out_dim is the plan's img size. */
int vpd_plan_stage_crops(vpd_plan_t* plan, const unsigned char* rgb_u8, const unsigned char* flow_u8,
                         const unsigned char* mask_u8, const float* noise, const vpd_aug_params* params, int n,
                         int height, int width, const float* mean_std6, float noise_sd, float* scratch,
                         void* workspace, void* stream);

/* Inference views of decoded u8 frames (apply_vpd_model.py:94-118 builds them through FrameDataset,
 * vpd_dataset/single_frame.py:377-400): view 0 = the frame, view 1 (k_views == 2) = its horizontal flip with the x-flow
 * negated; normalised like vpd_plan_stage_crops with identity parameters (bit-identical), no resize (height / width must be
 * the plan's), written to the stem staging buffer as n_frames * k_views crops in the order [f0 v0, f0 v1, f1 v0, ...].
 * Follow with vpd_forward_eval / vpd_graph_capture_eval with x == NULL. */
int vpd_plan_stage_views(vpd_plan_t* plan, const unsigned char* rgb_u8, const unsigned char* flow_u8, int n_frames,
                         int k_views, int height, int width, const float* mean_std6, void* workspace, void* stream);

int vpd_graph_capture_eval(vpd_plan_t* plan, const float* params, const float* x, int n, float* emb_out,
                           void* workspace, void* stream);
int vpd_graph_launch_eval(vpd_plan_t* plan, int n, void* stream);

/* The fused BatchNorm-backward launches synchronise their (fully resident) grid in the launch; every wait is bounded
 * (1 s) and a time-out is counted in a sticky word of the workspace instead of hanging the GPU.  Copies that count to
 * the host (synchronises `stream`); non-zero means the results of the step(s) since vpd_plan_init_workspace are not
 * to be trusted.  No reference counterpart (torch launches one kernel per BatchNorm pass). */
int vpd_plan_sync_errors(vpd_plan_t* plan, void* workspace, void* stream, unsigned* count_out);

/* Per-kernel-class timing for the roofline report (bench.py): when enabled, every conv launch is
 * bracketed by HIP events on its own stream.  One class per kernel family: 0 conv3x3_c64_persistent_kernel<224>,
 * 1 conv3x3_pws_kernel<256,128,352>, 2 conv3x3_pws_kernel<256,64,416> and <128,128,288>, 3 conv3x3_pws_kernel<128,64,288>
 * (conv3x3_ws_kernel, their one-tile-per-block twin, with VPD_PWS=0), 4 conv_igemm_kernel (gather), the ring GEMM conv1x1_ws_kernel
 * and the streaming 1x1 kernels, 5 conv_wgrad128_persistent_kernel / conv_wgrad_halo_grouped_kernel (stride-1 3x3, per stage, without the slab reduce),
 * 6 per-conv weight-gradient launches (stride-2 3x3 on conv_wgrad_halo_kernel, 1x1 on conv_wgrad_kernel),
 * 7 conv_stem_persistent_kernel.  vpd_plan_read_timing (nclasses >= 8) waits for the events, writes out[3*cls + {0,1,2}] = {launches, milliseconds, algorithmic FLOPs} and clears. */
int vpd_plan_set_timing(vpd_plan_t* plan, int enable);
int vpd_plan_read_timing(vpd_plan_t* plan, double* out, int nclasses);

/* ---- single-operator entry points (used by the parity tests; same kernels) ---- */
/* One implicit-GEMM conv launch on padded-NHWC bf16 tensors (forward conv or data-gradient conv).
 * tapset9 = {nr, nc, dy0, dys, dx0, dxs, w0, wrs, wcs}: tap (ir,ic) gathers input pixel
 * (y*istr + dy0 + ir*dys, x*istr + dx0 + ic*dxs) in padded coordinates and uses weight slice
 * w0 + ir*wrs + ic*wcs of w_bf16 [slice][Co][Kc].  stats (optional, pre-zeroed): f64 [16][2][Co] accumulator
 * rows (block b adds its per-channel sum / sum of squares into row b % 16 with fp64 atomics). */
int vpd_op_conv2d(const void* x_bf16, const void* w_bf16, void* y_bf16, double* stats, int n, int xHp, int xWp,
                  int xC, int yHp, int yWp, int yC, int ypad, int Hs, int Ws, int osub, int oph, int opw, int istr,
                  int Kc, int Co, const int* tapset9, int accumulate, void* stream);
int vpd_op_conv_bm(int M, int Co);
/* The same launch with the epilogues the plan uses besides store / statistics (stride-1 output grid, osub 1):
 *  - eval (models/module.py:41-47 folded: BatchNorm's running statistics as scale / shift [Co] floats, the block's identity
 *    path, ReLU): y = relu?(ep_scale * conv + ep_shift (+ res)), res = bf16 NHWC padded by 1 with Co channels or null;
 *  - accumulate onto a dense y (a data gradient on top of the identity path), the OLD value first multiplied by the ReLU
 *    bit map acc_mask [M][Co/8] (bit j of byte (m, c8) keeps channel 8 c8 + j) when given. */
int vpd_op_conv2d_ep(const void* x_bf16, const void* w_bf16, void* y_bf16, int n, int xHp, int xWp, int xC, int yHp, int yWp,
                     int yC, int ypad, int Hs, int Ws, int istr, int Kc, int Co, const int* tapset9, const float* ep_scale,
                     const float* ep_shift, const void* res_padded_bf16, int ep_relu, int accumulate,
                     const unsigned char* acc_mask, void* stream);
/* The same launch as a DATA GRADIENT that also takes the sums of the BatchNorm backward consuming its output d (epilogue
 * modes 6 / 7, reference: the reductions inside torch's batch_norm backward for models/module.py:41-43): g = d * mask with
 * mask = the ReLU bit map [M][Co/8] of that BatchNorm's activation, rows f64 [4][2][Co] (pre-zeroed) receive sum g and
 * sum g * z per channel (z: the BatchNorm's dense bf16 input [M][Co]).  y must be dense (ypad 0, yC == Co). */
int vpd_op_conv2d_bnsums(const void* x_bf16, const void* w_bf16, void* y_bf16, const void* bst_z_bf16,
                         const unsigned char* bst_mask, double* rows, int n, int xHp, int xWp, int xC, int Hs, int Ws,
                         int Kc, int Co, const int* tapset9, int accumulate, void* stream);
/* BatchNorm2d in training mode as the plan runs it (one launch: finalize + apply; models/module.py:41-43 + nn.BatchNorm2d):
 * rows f64 [4][2][C] hold the per-channel sum / sum of squares of z as the producing convolution's epilogue left them;
 * writes mean, rstd, scale = gamma rstd, shift = beta - mean scale, updates running_mean / running_var (momentum, unbiased
 * variance) when given, and out = relu?(scale z + shift (+ residual)) into the bf16 NHWC tensor padded by 1
 * ([n][H+2][W+2][C]; residual: same layout or null), plus the ReLU bit map [n H W][C/8] when mask_bits is given. */
int vpd_op_bn_forward(const void* z_bf16, const double* rows, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float* mean, float* rstd, float* scale, float* shift, const void* res_padded_bf16,
                      void* out_padded_bf16, unsigned char* mask_bits, int n, int H, int W, int C, int relu, float momentum,
                      float eps, void* stream);
/* BatchNorm backward, finalize + apply in one launch (bn_bwd_apply_fused_kernel): rows f64 [4][2][C] hold sum g and sum g * z
 * with g = dy * mask (what vpd_op_conv2d_bnsums leaves there); writes dz = gamma rstd (g - mean(g) - xhat mean(g xhat)) into
 * the bf16 NHWC tensor padded by 1, dgamma = sum g xhat, dbeta = sum g. */
int vpd_op_bn_backward_apply(const void* dy_bf16, const void* z_bf16, const unsigned char* mask_bits, const double* rows,
                             const float* gamma, const float* mean, const float* rstd, void* dz_padded_bf16, float* dgamma,
                             float* dbeta, int n, int H, int W, int C, void* stream);
/* dw[slice][Co][Kc] (fp32) += sum over output pixels of dz[m][co] * x[gather(m, tap)][kc].
 * slab: optional fp32 scratch of vpd_op_wgrad_slab_bytes() bytes; when given, eligible 3x3 stride-1 shapes use
 * the halo kernel (split partials in the slab + reduce), otherwise the generic kernel (fp32 atomics). */
int vpd_op_wgrad(const void* dz_bf16, const void* x_bf16, float* dw, int n, int dzHp, int dzWp, int dzC, int dzpad,
                 int xHp, int xWp, int xC, int Hs, int Ws, int istr, int Kc, int Co, const int* tapset9,
                 float* slab, void* stream);
size_t vpd_op_wgrad_slab_bytes(void);
/* dumps the ds_read_b64_tr_b16 fragments of one [128][64] bf16 tile: out [4][4][64][8] bf16 */
int vpd_op_tr_read_probe(const void* tile_bf16, void* out_bf16, void* stream);
/* Grouped weight gradients of `nprob` 3x3 pad-1 / 1x1 pad-0 convolutions (stride 1 or 2) on 128-channel-wide tiles in ONE persistent launch (the
 * form vpd_backward uses per ResNet stage; reference: the weight half of loss.backward(), models/util.py:52).
 * dims: 7 ints per problem {n, H, W, Co, Ci, stride, k} (H, W: output size; stride 1 or 2; k = 3: 3x3 pad 1, k = 1: 1x1 pad 0);
 * dz[i]: zero-bordered bf16 NHWC [n][H+2][W+2][Co]; x[i]: zero-bordered bf16 NHWC [n][stride*H+2][stride*W+2][Ci]; dw[i]: fp32
 * [k*k][Co][Ci]; slab[i]: vpd_op_wgrad128_slab_floats(Co, Ci) floats; dev_table: vpd_op_wgrad128_table_bytes() bytes. */
size_t vpd_op_wgrad128_table_bytes(void);
size_t vpd_op_wgrad128_slab_floats(int Co, int Ci);
int vpd_op_wgrad128_group(int nprob, const void* const* dz, const void* const* x, float* const* dw, float* const* slab,
                          const int* dims, void* dev_table, void* stream);
/* Host-only: the schedule vpd_op_wgrad128_group would build for `n` problems given as {M, Co, Ci, halo pixels} quadruples on
 * a device of G compute units (pixel split per problem, LPT deal of the (problem, tile, split) tasks to G persistent blocks).
 * Returns the number of tasks (-1: bad argument / more than `cap`); tasks: 4 ints each (problem, tile, split, 0). */
int vpd_op_wgrad128_schedule(int n, const int* dims4, int G, int* ksplit, int* blk_begin, int* tasks, int cap,
                             double* est_us);

#ifdef __cplusplus
}
#endif
#endif
