"""StudentEngine: device buffers + one libvpdhip plan for a student network.

PyTorch is used for device memory, streams and (in ddp.py) torch.distributed
only; every arithmetic step of the hot path is a hand-written HIP kernel behind
the C ABI (include/vpd_hip.h).  There is no fallback path.
"""
import ctypes as C
import os
from collections import OrderedDict

import torch

from ._lib import check, lib

# reference: models/module.py:17-32 (ENCODER_ARCH): BasicBlock archs and the Bottleneck archs
ENCODER_LAYERS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3),
                  "resnet101": (3, 4, 23, 3), "wide_resnet50_2": (3, 4, 6, 3), "wide_resnet101_2": (3, 4, 23, 3)}
BOTTLENECK_ARCHS = ("resnet50", "resnet101", "wide_resnet50_2", "wide_resnet101_2")
_STAGE_WIDTH = (64, 128, 256, 512)


def _world_size():
    """Ranks of the default process group (1 when torch.distributed is not initialised)."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def encoder_param_names(arch):
    """Trainable tensors / BN buffers of the encoder in reference state_dict order."""
    if arch not in ENCODER_LAYERS:
        raise KeyError("unsupported encoder_arch %r (%s)" % (arch, " | ".join(ENCODER_LAYERS)))
    train, bns = ["resnet.conv1.weight", "resnet.bn1.weight", "resnet.bn1.bias"], ["resnet.bn1"]
    inpl = 64
    exp = 4 if arch in BOTTLENECK_ARCHS else 1
    for li, (nblk, planes) in enumerate(zip(ENCODER_LAYERS[arch], _STAGE_WIDTH), start=1):
        for bi in range(nblk):
            p = "resnet.layer%d.%d" % (li, bi)
            stride = 2 if (bi == 0 and li > 1) else 1
            train += [p + ".conv1.weight", p + ".bn1.weight", p + ".bn1.bias",
                      p + ".conv2.weight", p + ".bn2.weight", p + ".bn2.bias"]
            bns += [p + ".bn1", p + ".bn2"]
            if exp == 4:
                train += [p + ".conv3.weight", p + ".bn3.weight", p + ".bn3.bias"]
                bns.append(p + ".bn3")
            if stride != 1 or inpl != planes * exp:
                train += [p + ".downsample.0.weight", p + ".downsample.1.weight", p + ".downsample.1.bias"]
                bns.append(p + ".downsample.1")
            inpl = planes * exp
    train += ["resnet.fc.weight", "resnet.fc.bias"]
    return train, bns


DECODER_PARAM_NAMES = ["layers.0.weight", "layers.0.bias", "layers.2.weight", "layers.2.bias",
                       "layers.5.weight", "layers.5.bias"]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _Plan:
    """One vpd_plan_t + its workspace for a fixed (H, W, max_batch, train, motion)."""

    def __init__(self, eng, h, w, max_batch, train, motion):
        L = self.L = eng.L
        check = self.check = eng.check
        self.h, self.w, self.max_batch, self.train, self.motion = h, w, max_batch, train, motion
        handle = C.c_void_p()
        # data parallel, VPD_DDP_EARLY_BUCKET0=1: bucket 0 (fc + layer4, 61 % of the gradient bytes) is handed to the reducer at the
        # end of layer4's backward instead of behind layer3's (include/vpd_hip.h, VPD_TRAIN_EARLY_BUCKET0).  OFF by default: the two
        # weight-gradient launches it takes cost 3.1 % of a rank's step (profiles/r05_ab_wg_unmerge.txt) for certain, what the earlier
        # all-reduce buys is unmeasured until a multi-GPU node runs bench.py with the switch both ways
        flags = int(bool(train))
        self.early_bucket0 = bool(train) and _world_size() > 1 and os.environ.get("VPD_DDP_EARLY_BUCKET0", "0") == "1"
        if self.early_bucket0:
            flags |= 2
        check(L.vpd_plan_create(eng.arch.encode(), eng.c_in, h, w, eng.emb_dim, int(motion), max_batch, flags,
                                C.byref(handle)), "vpd_plan_create")
        self.handle = handle
        # the C side is the source of truth for the flat layout: verify ours
        n = L.vpd_plan_num_tensors(handle)
        rows = []
        kind, dec, off, numel, ndim = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong(), C.c_int()
        dims = (C.c_int * 4)()
        for i in range(n):
            check(L.vpd_plan_tensor_info(handle, i, C.byref(kind), C.byref(dec), C.byref(off), C.byref(numel),
                                         C.byref(ndim), dims), "vpd_plan_tensor_info")
            rows.append((kind.value, dec.value, off.value, numel.value, tuple(dims[k] for k in range(ndim.value))))
        self.rows = rows
        self.param_numel = L.vpd_plan_param_numel(handle)
        self.ws_bytes = L.vpd_plan_workspace_bytes(handle)
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=eng.device)
        check(L.vpd_plan_init_workspace(handle, _ptr(self.workspace), eng._stream()), "vpd_plan_init_workspace")
        self.buckets = []
        o, m = C.c_longlong(), C.c_longlong()
        for b in range(L.vpd_plan_num_buckets(handle)):
            check(L.vpd_plan_bucket_range(handle, b, C.byref(o), C.byref(m)), "vpd_plan_bucket_range")
            self.buckets.append((o.value, m.value))
        self.packed_version = None
        self.graph_sizes = set()
        # lazy gradients under data parallelism: bucket b's conv weight gradients as a view of the workspace (the kernels' own
        # layout), and the flat-buffer positions of everything else (BatchNorm, fc, motion head, the stem conv)
        self.scratch_views, self.small_idx = [], None
        if train:
            for b in range(len(self.buckets)):
                check(L.vpd_plan_bucket_scratch_range(handle, b, C.byref(o), C.byref(m)), "vpd_plan_bucket_scratch_range")
                self.scratch_views.append(self.workspace[o.value:o.value + 4 * m.value].view(torch.float32))
            small = [(off_, n_) for i, (kind_, _, off_, n_, _) in enumerate(rows) if kind_ != 0 or i == 0]
            self.small_ranges = small
            # the C side is the source of truth: together the scratch views (every conv but the stem) and the small ranges
            # must cover the flat parameter buffer exactly once
            conv_numel = sum(n_ for i, (kind_, _, _, n_, _) in enumerate(rows) if kind_ == 0 and i != 0)
            assert sum(v.numel() for v in self.scratch_views) == conv_numel, "bucket scratch ranges != conv weight tensors"
            assert conv_numel + sum(n_ for _, n_ in small) == self.param_numel, "small ranges + conv tensors != param_numel"
            self.small_idx = torch.cat([torch.arange(a, a + n_, dtype=torch.int64) for a, n_ in small]).to(eng.device)

    def bn_table(self):
        L, check = self.L, self.check
        out = []
        ch, rm, rv = C.c_int(), C.c_longlong(), C.c_longlong()
        for i in range(L.vpd_plan_num_bn(self.handle)):
            check(L.vpd_plan_bn_info(self.handle, i, C.byref(ch), C.byref(rm), C.byref(rv)), "vpd_plan_bn_info")
            out.append((ch.value, rm.value, rv.value))
        return out

    def close(self):
        if self.handle:
            self.L.vpd_plan_destroy(self.handle)
            self.handle = None
        # whoever still holds this _Plan (apply's graph table, a reducer's last plan) must not pin its workspace
        self.workspace, self.scratch_views, self.small_idx = None, [], None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StudentEngine:
    """Flat fp32 parameter / gradient / optimizer-state buffers in the
    reference's state_dict order and layout, plus lazily created plans."""

    def __init__(self, arch, c_in, emb_dim, device="cuda", dtype="bf16"):
        if not torch.cuda.is_available():
            raise RuntimeError("vpd_amd needs a ROCm GPU (MI355X): torch.cuda.is_available() is False; "
                               "there is no CPU fallback for the student path")
        # dtype: element type of activations / packed weights -- "bf16" (libvpdhip.so: training and inference) or "fp16"
        # (libvpdhip_f16.so: inference only; the reference's own GPU precision, train_vpd_model.py:79)
        self.dtype = dtype
        self.loss_scale = 1.0      # set by models.util.LossScaler around a backward pass + optimizer step (fp16 training)
        self.L = lib(dtype)   # fail loudly right here if the HIP library is missing
        self.check = lambda rc, what="": check(rc, what, dtype)
        self.arch, self.c_in, self.emb_dim = arch, int(c_in), int(emb_dim)
        self.device = torch.device(device)
        self.enc_names, self.bn_names = encoder_param_names(arch)
        # a throw-away 1-crop plan tells us the flat layout (shapes/offsets) without duplicating it here
        probe = _Plan(self, 64, 64, 1, False, True)
        assert len(probe.rows) == len(self.enc_names) + len(DECODER_PARAM_NAMES)
        self.layout = OrderedDict()
        for name, row in zip(self.enc_names + ["decoder." + k for k in DECODER_PARAM_NAMES], probe.rows):
            self.layout[name] = row
        self.param_numel = probe.param_numel
        self.bn_layout = OrderedDict(zip(self.bn_names, probe.bn_table()))
        self.bn_numel = sum(2 * c for c, _, _ in self.bn_layout.values())
        probe.close()

        z = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.param_numel, **z)
        self._grads = torch.zeros(self.param_numel, **z)      # see the `grads` property
        self.bn_running = torch.zeros(max(self.bn_numel, 1), **z)
        # nn.BatchNorm2d.num_batches_tracked of every BatchNorm: counted on the host and added to the device tensor when it is
        # read (the property below, state_dict()) -- a one-element-per-BatchNorm add per step is a 5 us launch of its own
        self._nbt = torch.zeros(len(self.bn_names), dtype=torch.int64, device=self.device)
        self._nbt_pending = 0
        self.adam_m = None
        self.adam_v = None
        self.adam_step = 0
        self.loss_step = torch.zeros(1, **z)
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._plans = {}
        self._hip_version = 0          # bumped when a HIP kernel (AdamW) rewrites params behind torch's back
        self._last = None              # (plan, n) of the last train forward, consumed by backward
        self._step_plan = None         # plan of the last backward: its packed weights are refreshed by adamw_step
        self.bucket_events = None

    # -- helpers -------------------------------------------------------------
    @property
    def num_batches_tracked(self):
        """int64[len(bn_names)] on the device, up to date (train-mode forwards since the last read are added first)"""
        if self._nbt_pending:
            self._nbt += self._nbt_pending
            self._nbt_pending = 0
        return self._nbt

    @property
    def grads(self):
        """The flat fp32 gradient buffer (reference layouts; every p.grad is a view of it).  After a LAZY backward -- the one
        models.util.step() runs between forward and optimizer step, where the reference never looks at .grad either -- the
        conv weight gradients are still in the kernels' own layout: reading this property completes the buffer first."""
        self.materialize_grads()
        return self._grads

    def materialize_grads(self):
        pl = self._step_plan
        if pl is not None and pl.handle and self.L.vpd_plan_grads_pending(pl.handle):
            self.check(self.L.vpd_plan_materialize_grads(pl.handle, _ptr(self._grads), _ptr(pl.workspace), self._stream()),
                  "vpd_plan_materialize_grads")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def view(self, name, buf=None):
        kind, dec, off, numel, shape = self.layout[name]
        return (self.params if buf is None else buf)[off:off + numel].view(shape)

    def bn_views(self, name):
        c, rm, rv = self.bn_layout[name]
        return self.bn_running[rm:rm + c], self.bn_running[rv:rv + c]

    def weights_version(self):
        return (self.params._version, self.bn_running._version, self._hip_version)

    def mark_weights_changed(self):
        """Call after writing `params` / `bn_running` by a path torch's version counters do not see (a collective, a
        raw pointer): every plan repacks its bf16 weights before its next forward."""
        self._hip_version += 1

    def plan(self, h, w, n, train, motion):
        key = (h, w, bool(train), bool(motion))
        pl = self._plans.get(key)
        if pl is None or pl.max_batch < n:
            if pl is not None:
                if self._step_plan is pl:
                    self.materialize_grads()       # a pending lazy backward lives in the workspace that is about to go
                    self._step_plan = None
                pl.close()
            pl = _Plan(self, h, w, max(n, pl.max_batch if pl else 0), train, motion)
            assert pl.param_numel <= self.param_numel      # plans without the motion head omit its tensors
            self._plans[key] = pl
        return pl

    def _ensure_packed(self, pl):
        v = self.weights_version()
        if pl.packed_version != v:
            # the eval-mode BN fold is only needed by eval plans
            self.check(self.L.vpd_pack_weights(pl.handle, _ptr(self.params), None if pl.train else _ptr(self.bn_running),
                                         _ptr(pl.workspace), self._stream()), "vpd_pack_weights")
            pl.packed_version = v

    @staticmethod
    def _check_input(x, c_in):
        assert x.dim() == 4 and x.shape[1] == c_in, "expected f32 [N,%d,H,W], got %s" % (c_in, tuple(x.shape))
        assert x.dtype == torch.float32 and x.is_contiguous() and x.is_cuda

    # -- forward / backward / step ---------------------------------------------
    def forward_eval(self, x, target=None, motion=False, out=None, accumulate_loss=True, staged=None):
        if staged is not None:
            n, h = staged
            w = h
        else:
            self._check_input(x, self.c_in)
            n, _, h, w = x.shape
        pl = self.plan(h, w, n, False, motion)
        self._ensure_packed(pl)
        emb = out if out is not None else torch.empty((n, self.emb_dim), dtype=torch.float32, device=self.device)
        self.check(self.L.vpd_forward_eval(pl.handle, _ptr(self.params), _ptr(x), n, _ptr(emb), _ptr(target),
                                     _ptr(self.loss_step) if target is not None else None,
                                     _ptr(self.loss_accum) if (target is not None and accumulate_loss) else None,
                                     _ptr(pl.workspace), self._stream()), "vpd_forward_eval")
        return emb

    def stage_crops(self, rgb_u8, flow_u8, mask_u8, params_dev, noise, img_dim, mean_std6, noise_sd, scratch, train,
                    motion=False):
        """Device input pipeline (vpd_amd.augment) writing straight into the plan's stem staging buffer; follow with
        forward_train / forward_eval(x=None, staged=(n, img_dim))."""
        n, h, w, _ = rgb_u8.shape
        pl = self.plan(img_dim, img_dim, n, train, motion)
        ms = (C.c_float * 6)(*mean_std6)
        self.check(self.L.vpd_plan_stage_crops(pl.handle, _ptr(rgb_u8), _ptr(flow_u8), _ptr(mask_u8), _ptr(noise),
                                         _ptr(params_dev), n, h, w, ms, float(noise_sd), _ptr(scratch),
                                         _ptr(pl.workspace), self._stream()), "vpd_plan_stage_crops")
        return pl

    def stage_views(self, rgb_u8, flow_u8, k_views, mean_std6):
        """Inference views [frame, h-flip] of decoded u8 frames straight into the EVAL plan's stem staging buffer
        (vpd_plan_stage_views: table-driven normalisation, bit-identical to stage_crops with identity parameters)."""
        n, h, w, _ = rgb_u8.shape
        pl = self.plan(h, w, n * k_views, False, False)
        ms = (C.c_float * 6)(*mean_std6)
        self.check(self.L.vpd_plan_stage_views(pl.handle, _ptr(rgb_u8), _ptr(flow_u8), n, k_views, h, w, ms,
                                         _ptr(pl.workspace), self._stream()), "vpd_plan_stage_views")
        return pl

    def forward_train(self, x, target=None, motion=False, accumulate_loss=True, staged=None):
        if staged is not None:      # (n, img_dim): the batch is already in the staging buffer (stage_crops)
            n, h = staged
            w = h
        else:
            self._check_input(x, self.c_in)
            n, _, h, w = x.shape
        pl = self.plan(h, w, n, True, motion)
        self._ensure_packed(pl)
        emb = torch.empty((n, self.emb_dim), dtype=torch.float32, device=self.device)
        if target is not None:
            want = (n, self.emb_dim * (2 if motion else 1))
            assert tuple(target.shape) == want and target.dtype == torch.float32 and target.is_contiguous(), \
                "target must be f32 %s" % (want,)
        self.check(self.L.vpd_forward_train(pl.handle, _ptr(self.params), _ptr(self.bn_running), _ptr(x), _ptr(target), n,
                                      _ptr(emb), _ptr(self.loss_step),
                                      _ptr(self.loss_accum) if accumulate_loss else None,
                                      _ptr(pl.workspace), self._stream()), "vpd_forward_train")
        # BN running stats were rewritten by HIP kernels: packed eval scale/shift are stale
        self._hip_version += 1
        pl.packed_version = (self.params._version, self.bn_running._version, self._hip_version)
        if n > 0:
            self._nbt_pending += 1
        self._last = (pl, n) if target is not None else None
        return emb

    def backward(self, events=None, lazy=False):
        """lazy: leave the conv weight gradients in the weight-gradient kernels' scratch layout for the fused optimizer step
        (with bucket events, i.e. under data parallelism, the reducer must then be told: reduce(plan, lazy=True))."""
        if self._last is None:
            raise RuntimeError("backward() without a preceding train-mode forward with a target")
        pl, n = self._last
        self._last = None
        self.materialize_grads()           # a pending lazy backward of the PREVIOUS step plan (possibly another plan) first
        self._step_plan = pl
        ev = None
        if events is not None:
            ev = (C.c_void_p * len(events))(*[C.c_void_p(e) for e in events])
        if self.loss_scale != 1.0 or hasattr(self.L, "vpd_plan_set_loss_scale"):      # (absent only in an older A/B library)
            self.check(self.L.vpd_plan_set_loss_scale(pl.handle, float(self.loss_scale)), "vpd_plan_set_loss_scale")
        if lazy:      # (with bucket events the reducer sums the scratch ranges: GradBucketReducer.reduce(plan, lazy=True))
            self.check(self.L.vpd_plan_set_lazy_grads(pl.handle, 1), "vpd_plan_set_lazy_grads")
        self.check(self.L.vpd_backward(pl.handle, _ptr(self.params), _ptr(self._grads), n, ev, _ptr(pl.workspace),
                                 self._stream()), "vpd_backward")
        return pl

    def unscale_grads_(self):
        """LossScaler.step() for an optimizer that reads p.grad: the flat gradient buffer (completed first) x 1 / loss scale."""
        if self.loss_scale != 1.0:
            self.materialize_grads()
            self._grads.mul_(1.0 / self.loss_scale)
            if self._step_plan is not None:
                self.check(self.L.vpd_plan_set_loss_scale(self._step_plan.handle, 1.0), "vpd_plan_set_loss_scale")
            self.loss_scale = 1.0

    @property
    def encoder_numel(self):
        """Elements of the flat buffers that belong to the encoder (a multiple of 4: the decoder's tensors follow)."""
        off = self.layout["decoder." + DECODER_PARAM_NAMES[0]][2]
        return (off + 3) & ~3

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, numel=None):
        """numel: leading elements of the flat buffers to update (default: all).  An optimizer built without the motion head's
        parameters passes encoder_numel, as torch.optim.AdamW never touches tensors it was not given."""
        numel = self.param_numel if numel is None else int(numel)
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
        self.adam_step += 1
        pl = self._step_plan
        if pl is not None and numel < pl.param_numel:
            # an optimizer that was NOT given the motion head, behind a plan that trains it: the fused pass would update the
            # head's tensors too (torch.optim.AdamW never touches tensors it was not given) -- plain flat update of the leading
            # `numel` elements instead; the plan repacks its bf16 weights before its next forward
            self.materialize_grads()
            pl = None
        if pl is not None and pl.packed_version == self.weights_version() and os.environ.get("VPD_FUSED_ADAMW", "1") != "0":
            # the train plan of the last backward: AdamW + refresh of its packed bf16 weights in one pass
            self.check(self.L.vpd_plan_adamw_step(pl.handle, _ptr(self.params), _ptr(self._grads), _ptr(self.adam_m),
                                            _ptr(self.adam_v), max(numel, pl.param_numel), lr, betas[0], betas[1], eps,
                                            weight_decay, self.adam_step, _ptr(pl.workspace), self._stream()),
                  "vpd_plan_adamw_step")
            self._hip_version += 1
            pl.packed_version = self.weights_version()
        else:
            self.unscale_grads_()      # (the flat kernel has no plan to ask for the loss scale)
            self.check(self.L.vpd_adamw_step(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v),
                                       numel, lr, betas[0], betas[1], eps, weight_decay, self.adam_step,
                                       self._stream()), "vpd_adamw_step")
            self._hip_version += 1

    def capture_eval_graph(self, x, out):
        """hipGraph of the eval forward for this batch size, bound to x / out (capture needs a
        non-default stream; the graph itself is launched on the current stream)."""
        self._check_input(x, self.c_in)
        n, _, h, w = x.shape
        pl = self.plan(h, w, n, False, False)
        self._ensure_packed(pl)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        self.check(self.L.vpd_graph_capture_eval(pl.handle, _ptr(self.params), _ptr(x), n, _ptr(out), _ptr(pl.workspace),
                                           C.c_void_p(side.cuda_stream)), "vpd_graph_capture_eval")
        torch.cuda.current_stream(self.device).wait_stream(side)
        pl.graph_sizes.add(n)
        return pl

    def capture_eval_graph_staged(self, n, img_dim, out):
        """hipGraph of the eval forward for n crops whose input is ALREADY in the plan's stem staging buffer (stage_crops /
        CropAugmenter.stage_views): the graph starts at the stem convolution."""
        pl = self.plan(img_dim, img_dim, n, False, False)
        self._ensure_packed(pl)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        self.check(self.L.vpd_graph_capture_eval(pl.handle, _ptr(self.params), None, n, _ptr(out), _ptr(pl.workspace),
                                           C.c_void_p(side.cuda_stream)), "vpd_graph_capture_eval")
        torch.cuda.current_stream(self.device).wait_stream(side)
        pl.graph_sizes.add(n)
        return pl

    def sync_errors(self):
        """Grid-barrier time-outs counted by the train plans since their workspaces were initialised (host sync).
        Non-zero = a fused BatchNorm launch gave up waiting for its grid: results are not to be trusted."""
        total = 0
        out = C.c_uint(0)
        for pl in self._plans.values():
            if pl.train and pl.handle:
                self.check(self.L.vpd_plan_sync_errors(pl.handle, _ptr(pl.workspace), self._stream(), C.byref(out)),
                      "vpd_plan_sync_errors")
                total += out.value
        return total

    def set_timing(self, pl, enable):
        self.check(self.L.vpd_plan_set_timing(pl.handle, int(enable)), "vpd_plan_set_timing")

    def read_timing(self, pl):
        out = (C.c_double * 24)()
        self.check(self.L.vpd_plan_read_timing(pl.handle, out, 8), "vpd_plan_read_timing")
        # (class 1 -> conv3x3_pws_kernel<256,128,352> only with > 1 tile per block, else the class-6 tile; the non-persistent
        #  conv3x3_ws_kernel twins of every class remain behind VPD_PWS=0)
        names = ["conv3x3_c64_persistent_kernel<224>", "conv3x3_pws_kernel<256,128,352>", "conv3x3_pws_kernel<256,64,416> | <128,128,288>",
                 "conv3x3_pws_kernel<128,64,288>", "conv1x1_ws_kernel (stride-2 / 1x1 ring GEMM) + conv1x1_stream_kernel (Bottleneck 1x1, streaming / recompute) + conv_igemm_kernel (gather)", "conv_wgrad128_persistent_kernel (layer2-4) + conv_wgrad_halo_grouped_kernel (layer1)",
                 "conv_wgrad_halo_kernel<10|13> (stride 2) + conv_wgrad_kernel (1x1)",
                 "conv_stem_persistent_kernel<160> + conv_wgrad_stem_kernel"]
        return {names[i]: dict(launches=out[3 * i], ms=out[3 * i + 1], flops=out[3 * i + 2]) for i in range(8)}

    def launch_eval_graph(self, pl, n):
        self._ensure_packed(pl)
        self.check(self.L.vpd_graph_launch_eval(pl.handle, n, self._stream()), "vpd_graph_launch_eval")
