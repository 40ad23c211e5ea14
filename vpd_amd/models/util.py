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
    # same three calls as the reference.  The fused loss object offers a backward that leaves the conv weight gradients in
    # the kernels' own layout -- ONLY an optimizer that reads that layout may get it (FusedAdamW: `consumes_lazy_grads`).
    # Any other optimizer (the reference's step() accepts any: torch.optim.AdamW over encoder.parameters(), a wrapper that
    # looks at p.grad) gets the plain loss.backward(), which completes every p.grad view of the flat gradient buffer.
    lazy = _LAZY and getattr(optimizer, "consumes_lazy_grads", False) and hasattr(loss, "backward_for_step")
    (loss.backward_for_step if lazy else loss.backward)()
    optimizer.step()
    optimizer.zero_grad()
