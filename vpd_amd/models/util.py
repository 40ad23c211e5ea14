"""step(): same call sequence as reference models/util.py:50-58; LossScaler: the GradScaler of the fp16 build."""
import os

_LAZY = os.environ.get("VPD_LAZY_GRADS", "1") != "0"      # A/B switch: 0 = step() runs the plain loss.backward()


class LossScaler:
    """What torch.cuda.amp.GradScaler is to the reference's CUDA path (train_vpd_model.py:105; models/util.py:55-57) for a student
    built with dtype="fp16": `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`.

    The scale is STATIC (a power of two, default 256): activation gradients are stored in fp16 (6e-5 smallest normal), the loss is a
    SUM over crops (train_vpd_model.py:87) so d(loss)/d(emb) is O(1) per element whatever the batch, and weight gradients, BatchNorm
    sums and AdamW are fp32 -- 256 keeps the stem's activation gradients normal with 2^8 of head-room below fp16's 65,504.
    scale(loss).backward() leaves every .grad scaled, as GradScaler does; step() un-scales: the fused AdamW reads gradients x 1 / scale
    in its kernel, any other optimizer gets the flat gradient buffer multiplied by 1 / scale first.  A non-finite epoch loss is
    reported by ModelTrainer.epoch (there is no per-step inf check: it would cost a host sync per step)."""

    def __init__(self, engine, init_scale=256.0):
        if float(init_scale) <= 0:
            raise ValueError("loss scale must be positive")
        self._engine = engine
        self._scale = float(init_scale)

    def get_scale(self):
        return self._scale

    def scale(self, loss):
        if not hasattr(loss, "_t"):
            raise TypeError("LossScaler.scale() takes the loss object of ModelTrainer's forward")
        self._engine.loss_scale = self._scale
        return loss

    def step(self, optimizer):
        eng = self._engine
        if getattr(optimizer, "consumes_lazy_grads", False):
            optimizer.step()                       # (FusedAdamW: engine.adamw_step un-scales in the kernel)
        else:
            eng.unscale_grads_()
            optimizer.step()
        eng.loss_scale = 1.0

    def update(self):
        return None


def step(optimizer, scaler, loss):
    """The reference's step() (models/util.py:50-58): loss.backward(); optimizer.step() -- or, with a scaler,
    scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update() -- then optimizer.zero_grad().

    The bf16 build computes with fp32 accumulation and needs no loss scaling: get_optimizer() returns scaler=None there; a student
    built with dtype="fp16" (the reference's own GPU precision) gets a LossScaler."""
    if scaler is not None and not isinstance(scaler, LossScaler):
        raise ValueError("the HIP path takes scaler=None (bf16) or a vpd_amd.models.util.LossScaler (fp16), not %r" % type(scaler))
    # The fused loss object offers a backward that leaves the conv weight gradients in the kernels' own layout -- ONLY an
    # optimizer that reads that layout may get it (FusedAdamW: `consumes_lazy_grads`).  Any other optimizer (the reference's
    # step() accepts any: torch.optim.AdamW over encoder.parameters(), a wrapper that looks at p.grad) gets the plain
    # loss.backward(), which completes every p.grad view of the flat gradient buffer.
    lazy = _LAZY and getattr(optimizer, "consumes_lazy_grads", False) and hasattr(loss, "backward_for_step")
    if scaler is None:
        (loss.backward_for_step if lazy else loss.backward)()
        optimizer.step()
    else:
        scaled = scaler.scale(loss)
        (scaled.backward_for_step if lazy else scaled.backward)()
        scaler.step(optimizer)
        scaler.update()
    optimizer.zero_grad()
