"""step(): same call sequence as reference models/util.py:50-58."""
import os

_LAZY = os.environ.get("VPD_LAZY_GRADS", "1") != "0"      # A/B switch: 0 = step() runs the plain loss.backward()


def step(optimizer, scaler, loss):
    """loss.backward(); optimizer.step(); optimizer.zero_grad().

    `scaler` exists for signature compatibility: the reference uses fp16 autocast +
    GradScaler on CUDA; this build computes in bf16 with fp32 accumulation and
    needs no loss scaling, so get_optimizer() returns scaler=None."""
    if scaler is not None:
        raise ValueError("the bf16 HIP path does not use a GradScaler; pass scaler=None")
    # same three calls as the reference; the fused loss object offers a backward that hands the gradients to the optimizer
    # in the kernels' own layout (the reference's loop never reads .grad between these calls; loss.backward() does fill it)
    (getattr(loss, "backward_for_step", loss.backward) if _LAZY else loss.backward)()
    optimizer.step()
    optimizer.zero_grad()
