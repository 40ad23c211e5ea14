"""Parity at BASELINE.json's full size (ResNet-34, 5x128x128, D=128, 256 crops -- too big for the CPU oracle in a test)
through size-independent properties of the path:

* eval forward is per-crop: embeddings of a 256-crop batch == embeddings of its 100 + 156 split, and == the hipGraph launch;
* train step is permutation-invariant: shuffling the crops (and targets) of the batch leaves the loss, the BatchNorm
  batch statistics and every gradient unchanged, and permutes the embeddings;
* the sum-MSE loss is additive in the TARGET error for fixed embeddings: loss(t) recomputed on the host from the
  returned embeddings equals the device loss;
* fused AdamW over the whole 21.4 M-element flat buffer == torch.optim.AdamW stepping the same tensor on the device.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ARCH, C_IN, D, HW, B = "resnet34", 5, 128, 128, 256


def _model(seed=0, damp_residual=False):
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    enc = RGBF_EmbeddingModel(ARCH, D, True, "cuda")
    enc.reset_parameters(seed=seed)
    if damp_residual:      # the well-conditioned regime of tests/test_model_gpu.py (no chaotic amplification of rounding)
        with torch.no_grad():
            for name, p in enc.named_parameters():
                if name.endswith(".bn2.weight"):
                    p.mul_(0.1)
    return enc


def _batch(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    img = torch.randn((B, C_IN, HW, HW), generator=g, device="cuda")
    tgt = torch.randn((B, D), generator=g, device="cuda")
    return img, tgt


def test_eval_forward_is_per_crop_and_graph_exact():
    enc = _model()
    eng = enc.engine
    img, _ = _batch(1)
    enc.eval()
    full = eng.forward_eval(img).clone()
    a = eng.forward_eval(img[:100].contiguous()).clone()
    b = eng.forward_eval(img[100:].contiguous()).clone()
    assert torch.equal(full, torch.cat([a, b]))                # same per-pixel K order whatever the tiling
    out = torch.empty_like(full)
    pl = eng.capture_eval_graph(img, out)
    eng.launch_eval_graph(pl, B)
    torch.cuda.synchronize()
    assert torch.equal(out, full)
    assert torch.isfinite(full).all() and float(full.abs().mean()) > 1e-3


def test_train_step_is_permutation_invariant():
    from vpd_amd.trainer import ModelTrainer
    img, tgt = _batch(2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    res = []
    for p in (None, perm):
        enc = _model(damp_residual=True)
        tr = ModelTrainer(enc, False)
        enc.train()
        x, t = (img, tgt) if p is None else (img[p].contiguous(), tgt[p].contiguous())
        emb = enc.engine.forward_train(x, t, accumulate_loss=False).clone()
        loss = float(enc.engine.loss_step.item())
        enc.engine.backward()
        torch.cuda.synchronize()
        sd = enc.state_dict()
        res.append((emb, loss, enc.engine.grads.clone(), sd["resnet.layer3.2.bn1.running_mean"].clone(),
                    sd["resnet.bn1.running_var"].clone()))
    (e0, l0, g0, rm0, rv0), (e1, l1, g1, rm1, rv1) = res
    # the only differences are fp32 summation order (statistics, atomics) and the bf16 roundings they may flip
    assert abs(l0 - l1) <= 2e-3 * l0, (l0, l1)
    rl2 = lambda a, b: float((a - b).norm() / b.norm())
    assert rl2(rm0, rm1) <= 1e-2 and rl2(rv0, rv1) <= 1e-3, (rl2(rm0, rm1), rl2(rv0, rv1))
    rel_e = float((e1 - e0[perm]).norm() / e0.norm())
    rel_g = float((g1 - g0).norm() / g0.norm())
    # (at the plain initialisation the same comparison gives 2e-2 / 0.38: the summation-order difference of the
    #  statistics alone is amplified by the chaotic early-layer gradients described in tests/test_model_gpu.py)
    assert rel_e <= 1e-2 and rel_g <= 0.1, (rel_e, rel_g)
    # additivity of the sum-MSE: the device loss is the host's sum over crops of |e - t|^2
    host = float(((e0.double() - tgt.double()) ** 2).sum())
    assert abs(host - l0) <= 1e-4 * l0, (host, l0)


def test_fused_adamw_full_buffer_matches_torch():
    enc = _model()
    eng = enc.engine
    n = eng.param_numel
    g = torch.Generator(device="cuda").manual_seed(5)
    ref = torch.nn.Parameter(eng.params.detach().clone())
    opt = torch.optim.AdamW([ref], lr=5e-4)                    # torch defaults, as train_vpd_model.py:104
    for step in range(3):
        grad = torch.randn(n, generator=g, device="cuda") * (10.0 ** (step - 1))
        eng.grads.copy_(grad)
        eng.adamw_step(5e-4)
        ref.grad = grad.clone()
        opt.step()
        torch.cuda.synchronize()
        d = (eng.params - ref.detach()).abs()
        assert float(d.max()) <= 2e-6, (step, float(d.max()))
    st = opt.state[ref]
    assert torch.allclose(eng.adam_m, st["exp_avg"], rtol=1e-5, atol=1e-8)
    assert torch.allclose(eng.adam_v, st["exp_avg_sq"], rtol=1e-5, atol=1e-10)
    # checksum of checksums: per-bucket sums of the updated parameters agree
    pl = eng.plan(HW, HW, 8, True, False)
    for off, numel in pl.buckets:
        a, b = float(eng.params[off:off + numel].double().sum()), float(ref.detach()[off:off + numel].double().sum())
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)) + 1e-3
