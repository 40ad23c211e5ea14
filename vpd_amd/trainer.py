"""ModelTrainer: the reference's training driver (train_vpd_model.py:53-112)
on the HIP engine.  Same methods, same epoch() return value (sum of per-batch
sum-MSE / number of crops), same checkpoint files."""
import os

import torch

from .ddp import GradBucketReducer
from .models.module import FCNet
from .models.util import LossScaler, step


class _Loss:
    """What `F.mse_loss(...)` returns in the reference loop, for the fused path:
    supports .backward() (models/util.py:52) and .item() (train_vpd_model.py:93)."""

    def __init__(self, trainer):
        self._t = trainer

    def backward(self):
        self._t._backward()

    def backward_for_step(self):
        """backward() as models.util.step() runs it: the optimizer step follows immediately and nothing reads .grad in
        between (reference models/util.py:52-58), so the conv weight gradients may stay in the kernels' layout."""
        self._t._backward(lazy=True)

    def item(self):
        return float(self._t.encoder.engine.loss_step.item())   # host sync, as in the reference


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW(params, lr) with torch defaults (train_vpd_model.py:104) as ONE
    HIP kernel over the engine's flat fp32 buffers (weight decay on every tensor)."""

    consumes_lazy_grads = True      # models.util.step(): this optimizer reads the weight-gradient kernels' own layout

    def __init__(self, params, engine, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._engine = engine
        # like torch.optim.AdamW, touch only the tensors that were passed in: without the motion head's parameters the
        # update stops where the encoder's tensors end in the flat buffers
        enc_end = engine.encoder_numel
        ptr0 = engine.params.data_ptr()
        has_dec = any((q.data_ptr() - ptr0) // 4 >= enc_end for q in params)
        self._numel = engine.param_numel if has_dec else enc_end

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self._engine.adamw_step(g["lr"], g["betas"], g["eps"], g["weight_decay"], numel=self._numel)

    def zero_grad(self, set_to_none=False):
        # gradients live in the engine's flat buffer and are overwritten (not accumulated)
        # by the next backward, which is what zero_grad-after-every-step amounts to
        return None


class ModelTrainer:
    """Class for training the encoder. Discarded after training"""

    def __init__(self, encoder, motion, process_group=None, augmenter=None, augment=True):
        """augmenter: a vpd_amd.augment.CropAugmenter -> epoch() also accepts RAW batches
        {'rgb_u8': u8[B,H,W,3], 'flow_u8': u8[B,H,W,2], 'mask_u8': u8[B,H,W] (optional), 'flip': int[B] (optional,
        decided by the dataset because it selects the teacher row), 'emb'} and runs the reference's per-item
        transforms (vpd_dataset/single_frame.py:168-206) on the device, straight into the stem's staging buffer."""
        device = encoder.device
        self.augmenter, self.augment = augmenter, bool(augment)
        self.encoder = encoder.to(device)
        self.motion = bool(motion)
        if motion:
            self.fcn_time = FCNet(encoder.engine, encoder.emb_dim, [128, 128], 2 * encoder.emb_dim, dropout=0)
        self._reducer = None
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                         and torch.distributed.get_world_size() > 1):
            self._reducer = GradBucketReducer(encoder.engine, process_group)

    # -- pieces of the reference loop body (train_vpd_model.py:79-91) -------------------
    def _forward_loss(self, img, gt_emb, train):
        eng = self.encoder.engine
        img = img.to(eng.device, dtype=torch.float32, non_blocking=True).contiguous()
        gt = gt_emb.to(eng.device, dtype=torch.float32, non_blocking=True).contiguous()
        if train:
            eng.forward_train(img, gt, motion=self.motion)
        else:
            eng.forward_eval(img, gt, motion=self.motion)
        return _Loss(self)

    def _forward_loss_raw(self, batch, train):
        from .augment import sample_params
        if self.augmenter is None:
            raise RuntimeError("raw u8 batches need ModelTrainer(..., augmenter=CropAugmenter(...))")
        eng = self.encoder.engine
        dev = lambda t: None if t is None else t.to(eng.device, non_blocking=True).contiguous()
        rgb = dev(batch['rgb_u8'])
        n, h, w, _ = rgb.shape
        # like the reference validation batches are augmented too: augmentation is a property
        # of the dataset objects, which are all built with augment=True (vpd_dataset/single_frame.py:267-272)
        # seed=None: a fresh Philox key for the device-side mask noise of THIS batch (drawn from torch's global RNG)
        params = sample_params(n, h, w, augment=self.augment, flip=False)
        self.last_aug_params = params
        if 'flip' in batch:
            params['flip'] = batch['flip'].numpy() if hasattr(batch['flip'], 'numpy') else batch['flip']
        staged = self.augmenter.stage(eng, rgb, dev(batch.get('flow_u8')), dev(batch.get('mask_u8')), params,
                                      train=train, motion=self.motion)
        gt = batch['emb'].to(eng.device, dtype=torch.float32, non_blocking=True).contiguous()
        if train:
            eng.forward_train(None, gt, motion=self.motion, staged=staged)
        else:
            eng.forward_eval(None, gt, motion=self.motion, staged=staged)
        return _Loss(self)

    def _backward(self, lazy=False):
        eng = self.encoder.engine
        if self._reducer is not None:
            nb = len(eng._last[0].buckets) if eng._last is not None else 0      # (backward() raises without a forward)
            # lazy gradients stay on under data parallelism: a SUM all-reduce is layout-agnostic, so the reducer sums the
            # weight-gradient scratch ranges + the small tensors of the flat buffer (VPD_DDP_LAZY=0: flat buffer, eager unpack)
            lazy = lazy and os.environ.get("VPD_DDP_LAZY", "1") != "0"
            pl = eng.backward(self._reducer.event_handles(nb), lazy=lazy)
            self._reducer.reduce(pl, lazy=lazy)
        else:
            eng.backward(lazy=lazy)
        if not lazy:
            self._reattach_grads()

    def _reattach_grads(self):
        """A torch optimizer's zero_grad() drops .grad (set_to_none=True is torch's default): every plain backward() makes
        sure the parameters' .grad are the views of the engine's flat gradient buffer again."""
        eng = self.encoder.engine
        first = next(iter(self.encoder.parameters()))
        if first.grad is not None and (not hasattr(self, 'fcn_time') or next(iter(self.fcn_time.parameters())).grad is not None):
            return
        g = eng.grads
        for k in eng.enc_names:
            self.encoder.get_parameter(k).grad = eng.view(k, g)
        if hasattr(self, 'fcn_time'):
            from .models.module import DECODER_PARAM_NAMES
            for k in DECODER_PARAM_NAMES:
                self.fcn_time.get_parameter(k).grad = eng.view("decoder." + k, g)

    def epoch(self, data_loader, optimizer=None, scaler=None, progress_cb=None):
        eng = self.encoder.engine
        self.encoder.eval() if optimizer is None else self.encoder.train()
        if hasattr(self, 'fcn_time'):
            self.fcn_time.eval() if optimizer is None else self.fcn_time.train()

        eng.loss_accum.zero_()          # the epoch accumulator of train_vpd_model.py:73,93 lives on device
        epoch_emb_n = 0
        for batch in data_loader:
            if 'rgb_u8' in batch:
                n = batch['rgb_u8'].shape[0]
                loss = self._forward_loss_raw(batch, train=optimizer is not None)
            else:
                n = batch['img'].shape[0]
                loss = self._forward_loss(batch['img'], batch['emb'], train=optimizer is not None)
            if optimizer is not None:
                step(optimizer, scaler, loss)
            epoch_emb_n += n
            if progress_cb is not None:
                progress_cb(n)
        epoch_emb_loss = float(eng.loss_accum.item())   # ONE host sync per epoch
        if optimizer is not None and eng.sync_errors():
            raise RuntimeError("a fused BatchNorm launch timed out at its in-launch grid barrier (the grid was not "
                               "resident): this epoch's results are invalid; set VPD_FUSED_BN=0 to use separate launches")
        if optimizer is not None and scaler is not None and epoch_emb_loss != epoch_emb_loss:
            raise FloatingPointError("non-finite training loss under fp16 with loss scale %g: an activation gradient left fp16's range "
                                     "(the reference's GradScaler would have skipped those steps); use LossScaler(engine, init_scale=<smaller>)"
                                     % scaler.get_scale())
        if self._reducer is not None:
            epoch_emb_loss, epoch_emb_n = self._reducer.all_reduce_scalars(epoch_emb_loss, epoch_emb_n)
        return epoch_emb_loss / epoch_emb_n

    def get_optimizer(self, learning_rate):
        params = list(self.encoder.parameters())
        if hasattr(self, 'fcn_time'):
            params.extend(self.fcn_time.parameters())
        # bf16 operands with fp32 accumulation need no GradScaler: scaler None.  A student built with dtype="fp16" gets the static
        # LossScaler (the reference: GradScaler() on 'cuda', train_vpd_model.py:104-105)
        eng = self.encoder.engine
        return FusedAdamW(params, eng, lr=learning_rate), (LossScaler(eng) if eng.dtype == "fp16" else None)

    def save_model(self, save_dir, name):
        torch.save(self.encoder.state_dict(),
                   os.path.join(save_dir, '{}.encoder.pt'.format(name)))
        if hasattr(self, 'fcn_time'):
            torch.save(self.fcn_time.state_dict(),
                       os.path.join(save_dir, '{}.decoder.pt'.format(name)))
