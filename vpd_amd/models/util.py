"""step(): same call sequence as reference models/util.py:50-58."""


def step(optimizer, scaler, loss):
    """loss.backward(); optimizer.step(); optimizer.zero_grad().

    `scaler` exists for signature compatibility: the reference uses fp16 autocast +
    GradScaler on CUDA; this build computes in bf16 with fp32 accumulation and
    needs no loss scaling, so get_optimizer() returns scaler=None."""
    if scaler is not None:
        raise ValueError("the bf16 HIP path does not use a GradScaler; pass scaler=None")
    loss.backward()
    optimizer.step()
    optimizer.zero_grad()
