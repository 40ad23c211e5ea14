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


# ---------------------------------------------------------------------------
# BASELINE configs[2] / configs[3] at their per-GPU size: --motion two-stream head, 512 crops per GPU,
# figure-skating-shaped crops (RGB_MEAN_STD['fs']), and the ragged final batch of a 4096-crop global batch on 8 GPUs
# (3616 = 8 x 452) against the SAME plan
# ---------------------------------------------------------------------------
B4, RAGGED = 512, 452


def _fs_batch(n, seed, motion=True):
    from vpd_amd.data import RGB_MEAN_STD
    g = torch.Generator(device="cuda").manual_seed(seed)
    rgb = torch.randint(0, 256, (n, 3, HW, HW), generator=g, device="cuda").float() / 255.0
    mean = torch.tensor(RGB_MEAN_STD["fs"][0], device="cuda").view(1, 3, 1, 1)
    std = torch.tensor(RGB_MEAN_STD["fs"][1], device="cuda").view(1, 3, 1, 1)
    flow = (124 + 12 * torch.randn((n, 2, HW, HW), generator=g, device="cuda")).round().clamp(0, 255) / 255.0 - 0.5
    img = torch.cat([(rgb - mean) / std, flow], dim=1).contiguous()
    t = torch.randn((n, D), generator=g, device="cuda")
    tgt = torch.cat([t, t - torch.randn((n, D), generator=g, device="cuda")], dim=1).contiguous() if motion else t
    return img, tgt


def test_c3_c4_shape_512_crops_motion_fs_and_ragged_452():
    from vpd_amd.trainer import ModelTrainer
    img, tgt = _fs_batch(B4, 11)
    # --- eval forward: per-crop, so the 512-crop batch == its 256 + 256 split == the hipGraph launch, bit for bit;
    #     the ragged 452-crop batch through the same (512-crop) plan == the first 452 rows
    enc = _model()
    eng = enc.engine
    enc.eval()
    full = eng.forward_eval(img).clone()
    halves = torch.cat([eng.forward_eval(img[:256].contiguous()).clone(), eng.forward_eval(img[256:].contiguous()).clone()])
    assert torch.equal(full, halves)
    rag = eng.forward_eval(img[:RAGGED].contiguous()).clone()
    assert torch.equal(rag, full[:RAGGED])
    out = torch.empty_like(full)
    pl = eng.capture_eval_graph(img, out)
    eng.launch_eval_graph(pl, B4)
    torch.cuda.synchronize()
    assert torch.equal(out, full) and torch.isfinite(full).all()
    del enc, eng, pl
    # --- train step with the motion head: permutation invariance at 512 and at the ragged 452 (same plan: the
    #     second call reuses the 512-crop workspace), device loss == host sum over crops of |pred - target|^2
    perm = {n: torch.randperm(n, generator=torch.Generator().manual_seed(3 + n)).cuda() for n in (B4, RAGGED)}
    for n in (B4, RAGGED):
        res = []
        for p in (None, perm[n]):
            enc = _model(damp_residual=True)
            tr = ModelTrainer(enc, True)
            torch.manual_seed(5)
            with torch.no_grad():
                for q in tr.fcn_time.parameters():               # same motion head in both runs
                    q.copy_(torch.randn(q.shape, generator=torch.Generator().manual_seed(q.numel())).to(q.device) * 0.05)
            enc.train()
            enc.engine.plan(HW, HW, B4, True, True)              # the 512-crop plan serves the ragged batch too
            x, t = img[:n], tgt[:n]
            x, t = (x.contiguous(), t.contiguous()) if p is None else (x[p].contiguous(), t[p].contiguous())
            emb = enc.engine.forward_train(x, t, motion=True, accumulate_loss=False).clone()
            loss = float(enc.engine.loss_step.item())
            enc.engine.backward()
            torch.cuda.synchronize()
            assert enc.engine._step_plan.max_batch == B4
            res.append((emb, loss, enc.engine.grads.clone(), tr))
        (e0, l0, g0, tr0), (e1, l1, g1, _) = res
        assert abs(l0 - l1) <= 2e-3 * l0, (n, l0, l1)
        rel_e = float((e1 - e0[perm[n]]).norm() / e0.norm())
        rel_g = float((g1 - g0).norm() / g0.norm())
        assert rel_e <= 1e-2 and rel_g <= 0.1, (n, rel_e, rel_g)
        assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
        with torch.no_grad():                                    # FCNet D->128->128->2D (models/module.py:139-156)
            w = tr0.fcn_time.state_dict()
            lin = torch.nn.functional.linear
            pred = lin(torch.relu(lin(torch.relu(lin(e0, w["layers.0.weight"], w["layers.0.bias"])),
                                      w["layers.2.weight"], w["layers.2.bias"])), w["layers.5.weight"], w["layers.5.bias"])
        host = float(((pred.double() - tgt[:n].double()) ** 2).sum())
        assert abs(host - l0) <= 2e-3 * l0, (n, host, l0)


def test_c5_shape_1000_crop_graph_batch_equals_eager():
    """apply_vpd_model.py's batch (BATCH_SIZE 500 frames x 2 views = 1000 crops, apply_vpd_model.py:15) as ONE hipGraph
    launch: bit-identical to the eager eval forward of the same batch and to ten 100-crop calls; the tail batch of a
    1 M-crop run (fewer frames) gets its own graph on the same plan."""
    enc = _model()
    eng = enc.engine
    enc.eval()
    n = 1000
    g = torch.Generator(device="cuda").manual_seed(21)
    img = torch.randn((n, C_IN, HW, HW), generator=g, device="cuda")
    eager = eng.forward_eval(img).clone()
    out = torch.empty_like(eager)
    pl = eng.capture_eval_graph(img, out)
    eng.launch_eval_graph(pl, n)
    torch.cuda.synchronize()
    assert torch.equal(out, eager) and torch.isfinite(out).all()
    parts = torch.cat([eng.forward_eval(img[i:i + 100].contiguous()).clone() for i in range(0, n, 100)])
    assert torch.equal(parts, eager)
    tail = 2 * 157                                              # a Diving48-sized last video
    out_t = torch.empty((tail, D), dtype=torch.float32, device="cuda")
    xt = img[:tail].contiguous()
    pl2 = eng.capture_eval_graph(xt, out_t)
    eng.launch_eval_graph(pl2, tail)
    out.zero_()
    eng.launch_eval_graph(pl, n)                                # both graphs stay valid side by side
    torch.cuda.synchronize()
    assert pl2 is pl and torch.equal(out_t, eager[:tail]) and torch.equal(out, eager)


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


# ---------------------------------------------------------------------------
# Values the REFERENCE produced at full size: tests/golden/c2_* / c3_* / c5_*.npz come from
# oracle/gen_golden.py running the reference's own RGBF_EmbeddingModel / ModelTrainer / FCNet on the CPU at 256 / 512 /
# 1000 crops (train_vpd_model.py:77-98, models/rgb.py:72-86); the HIP path is held to the same EMB_TOL / LOSS_TOL as on
# the small golden cases of tests/test_model_gpu.py.
# ---------------------------------------------------------------------------
import json
import os

from oracle import vpd_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
EMB_TOL, LOSS_TOL = 2e-2, 1e-2
# TRAIN-mode embeddings (batch statistics): every conv output z is stored in bf16, its statistics are taken over the rounded
# values and the normalised activation is rounded again -- two roundings per layer where the eval forward (BatchNorm folded
# into the conv epilogue, applied to the fp32 accumulators) has one.  Measured with the oracle's emulate_bf16 mode (the same
# algorithm and rounding points on the CPU) against the reference's fp32 emb_train: per-sample 0.034 mean / 0.035 max at 8
# crops, 0.034 / 0.042 at 64 crops (ResNet-34, procedural weights); the HIP path measures 0.030-0.041 at 256 crops.
TRAIN_EMB_TOL = 6e-2


def _golden_build(meta):
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.trainer import ModelTrainer
    enc_sd = O.procedural_state_dict(O.encoder_schema(meta["arch"], meta["c_in"], meta["emb_dim"]), meta["seed"])
    dec_sd = O.procedural_state_dict(O.decoder_schema(meta["emb_dim"]), meta["seed"] + 7) if meta["motion"] else None
    img = O.synthetic_crops(meta["n"], meta["c_in"], meta["hw"], meta["seed"] + 1,
                            O.FS_MEAN_STD if meta.get("norm") == "fs" else None)
    tgt = O.synthetic_targets(meta["n"], meta["emb_dim"], meta["motion"], meta["seed"] + 2)
    enc = RGBF_EmbeddingModel(meta["arch"], meta["emb_dim"], meta["c_in"] != 3, "cuda",
                              in_channels=None if meta["c_in"] in (3, 5) else meta["c_in"])
    enc.load_state_dict(enc_sd)
    tr = ModelTrainer(enc, meta["motion"])
    if meta["motion"]:
        tr.fcn_time.load_state_dict(dec_sd)
    return enc, tr, img, tgt


def _per_sample_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("name", ["c2_r34_c5_d128_m0_n256", "c3_r34_c5_d128_m1_n512_fs", "c3_r34_c6_d128_m1_n512",
                                  "c5_r34_c5_d128_n1000"])
def test_fullsize_values_match_the_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    enc, tr, img, tgt = _golden_build(meta)
    rec = {"meta": meta}
    # eval-mode embed(): every crop against the reference's embedding of the same crop
    e = enc.embed(img.numpy())
    ps = _per_sample_rel(e, g["emb_eval"])
    rec["emb_eval_per_sample_max"] = float(ps.max())
    assert e.shape == g["emb_eval"].shape and ps.max() <= EMB_TOL, float(ps.max())
    if meta["level"] == "eval":
        # the apply path's own route: ONE hipGraph launch of the 1000-crop batch (apply_vpd_model.py:15, :160-162)
        x = img.cuda()
        out = torch.empty((meta["n"], meta["emb_dim"]), dtype=torch.float32, device="cuda")
        enc.eval()
        pl = enc.engine.capture_eval_graph(x, out)
        enc.engine.launch_eval_graph(pl, meta["n"])
        torch.cuda.synchronize()
        assert _per_sample_rel(out.cpu().numpy(), g["emb_eval"]).max() <= EMB_TOL
        return
    ev = tr.epoch([{"img": img, "emb": tgt}])
    assert abs(ev - float(g["epoch_eval"])) <= LOSS_TOL * abs(float(g["epoch_eval"])), (ev, float(g["epoch_eval"]))
    # train-mode forward: embeddings under batch statistics, loss, BatchNorm taps
    enc, tr, img, tgt = _golden_build(meta)
    enc.train()
    eng = enc.engine
    emb = eng.forward_train(img.cuda(), tgt.cuda(), motion=meta["motion"], accumulate_loss=False).clone()
    l_hip = float(eng.loss_step.item())
    rec["loss_train"] = [l_hip, float(g["loss_train"])]
    assert abs(l_hip - float(g["loss_train"])) <= LOSS_TOL * abs(float(g["loss_train"])), rec["loss_train"]
    ps = _per_sample_rel(emb.cpu().numpy(), g["emb_train"])
    rec["emb_train_per_sample_max"] = float(ps.max())
    assert ps.max() <= TRAIN_EMB_TOL, float(ps.max())
    if meta["level"] == "light":
        return
    # full: gradients (per-stage norms against the reference's recorded norms), running statistics after the step
    eng.backward()
    torch.cuda.synchronize()
    stage = lambda k: k.split(".")[2] if k.startswith("enc.resnet.layer") else ("stem" if "conv1" in k or "bn1" in k else "fc")
    hip2, ref2 = {}, {}
    for k in [f[len("gnorm/"):] for f in g.files if f.startswith("gnorm/")]:
        got = enc.get_parameter(k[4:]).grad
        s_ = stage(k)
        hip2[s_] = hip2.get(s_, 0.0) + float(got.double().pow(2).sum())
        ref2[s_] = ref2.get(s_, 0.0) + float(g["gnorm/" + k]) ** 2
    rec["grad_norm_ratio_by_stage"] = {s_: (hip2[s_] / ref2[s_]) ** 0.5 for s_ in ref2}
    # bf16 operands against an fp32 reference: the norm of a stage's gradient is held to 15 % (the per-tensor direction is
    # gated on the small cases and in test_backward_matches_bf16_emulation_directly, where an emulation can be afforded)
    assert all(0.85 <= v <= 1.15 for v in rec["grad_norm_ratio_by_stage"].values()), rec["grad_norm_ratio_by_stage"]
    # BatchNorm running statistics after ONE reference step (epoch_train ran on fresh modules: same single update)
    sd = enc.state_dict()
    worst = 0.0
    for f in g.files:
        if f.startswith("post/") and not f.endswith("num_batches_tracked"):
            worst = max(worst, _rel(sd[f[len("post/"):]].cpu().numpy(), g[f]))
    rec["running_stats_rel_l2_max"] = worst
    assert worst <= 2e-2, worst
    out_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_fullsize_%s.json" % name), "w") as fp:
        json.dump(rec, fp, indent=1)


_STEM_POOL_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r})
from vpd_amd.models.rgb import RGBF_EmbeddingModel
out = {{}}
for arch, cin, hw, n in (("resnet34", 5, 128, 1000), ("resnet34", 5, 128, 70), ("resnet18", 3, 64, 96), ("resnet18", 5, 128, 10)):
    enc = RGBF_EmbeddingModel(arch, 32, cin == 5, "cuda")
    enc.reset_parameters(seed=3)
    # non-trivial folded BatchNorm (negative scales too: the maximum must be taken AFTER scale / shift / ReLU)
    sd = enc.state_dict()
    g = torch.Generator().manual_seed(5)
    sd["resnet.bn1.weight"] = torch.randn(64, generator=g)
    sd["resnet.bn1.bias"] = torch.randn(64, generator=g) * 0.3
    sd["resnet.bn1.running_mean"] = torch.randn(64, generator=g) * 0.2
    sd["resnet.bn1.running_var"] = torch.rand(64, generator=g) + 0.5
    enc.load_state_dict(sd)
    enc.eval()
    x = torch.randn((n, cin, hw, hw), generator=torch.Generator().manual_seed(n), dtype=torch.float32).cuda()
    out["%s_%d_%d_%d" % (arch, cin, hw, n)] = enc.engine.forward_eval(x).cpu().numpy()
np.savez({out!r}, **out)
"""


def test_stem_pool_in_the_conv_epilogue_equals_the_two_launch_path(tmp_path):
    """Eval stem: conv + folded BatchNorm + ReLU + 3x3 stride-2 max-pool in ONE launch (conv_stem_persistent_kernel<160, 9>,
    ConvParams::pool_y) against conv -> stem_pool_pair_kernel (VPD_STEM_POOL_FUSED=0): same rounding points, so the embeddings
    must agree BIT FOR BIT -- at 1,000 crops (four images per block, the carried row between tiles), at 70 (fewer images than
    CUs), at 64-pixel inputs (four conv rows = two pooled rows per tile), and below the 64-image threshold (same path).
    Reference: models/module.py:112-116 (conv1 -> bn1 -> relu -> maxpool)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for flag in ("1", "0"):
        out = str(tmp_path / ("e%s.npz" % flag))
        r = subprocess.run([sys.executable, "-c", _STEM_POOL_SCRIPT.format(repo=repo, out=out)],
                           env=dict(os.environ, VPD_STEM_POOL_FUSED=flag), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        res.append(np.load(out))
    for k in res[0].files:
        a, b = res[0][k], res[1][k]
        assert np.isfinite(a).all() and np.abs(a).max() > 0, k
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))


# ---------------------------------------------------------------------------
# fp16 elements (libvpdhip_f16.so; apply_vpd_model.py --dtype fp16): the reference's own GPU precision (fp16 autocast,
# train_vpd_model.py:79) for inference.  11 significant bits instead of bf16's 8: the eval embeddings are held to 1.5e-3 per sample
# against the reference's fp32 CPU embeddings (bf16: 2e-2 gate, 3-7e-3 measured) -- at full size and on the small fixtures of every
# student family.
# ---------------------------------------------------------------------------
FP16_EMB_TOL = 1.5e-3


def _fp16_encoder(meta):
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    enc_sd = O.procedural_state_dict(O.encoder_schema(meta["arch"], meta["c_in"], meta["emb_dim"]), meta["seed"])
    img = O.synthetic_crops(meta["n"], meta["c_in"], meta["hw"], meta["seed"] + 1,
                            O.FS_MEAN_STD if meta.get("norm") == "fs" else None)
    enc = RGBF_EmbeddingModel(meta["arch"], meta["emb_dim"], meta["c_in"] != 3, "cuda",
                              in_channels=None if meta["c_in"] in (3, 5) else meta["c_in"], dtype="fp16")
    enc.load_state_dict(enc_sd)
    return enc, img


@pytest.mark.parametrize("name", ["c5_r34_c5_d128_n1000", "c2_r34_c5_d128_m0_n256", "c3_r34_c6_d128_m1_n512"])
def test_fp16_inference_matches_the_reference_at_full_size(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    enc, img = _fp16_encoder(meta)
    e = enc.embed(img.numpy())
    ps = _per_sample_rel(e, g["emb_eval"])
    out_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_fp16_%s.json" % name), "w") as fp:
        json.dump({"emb_eval_per_sample_max": float(ps.max()), "emb_eval_per_sample_mean": float(ps.mean()), "tol": FP16_EMB_TOL}, fp)
    assert e.shape == g["emb_eval"].shape and ps.max() <= FP16_EMB_TOL, float(ps.max())
    # the apply path's own route: ONE hipGraph launch == the eager forward, bit for bit
    x = img.cuda()
    out = torch.empty((meta["n"], meta["emb_dim"]), dtype=torch.float32, device="cuda")
    enc.eval()
    pl = enc.engine.capture_eval_graph(x, out)
    enc.engine.launch_eval_graph(pl, meta["n"])
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), e)


def test_fp16_inference_on_the_small_fixtures():
    import glob
    worst = {}
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*r[0-9]*_c[0-9]_*.npz"))):
        g = np.load(path)
        meta = json.loads(str(g["meta"]))
        if "emb_eval" not in g.files or meta.get("level") == "eval" or meta["n"] > 64:
            continue
        enc, img = _fp16_encoder(meta)
        ps = _per_sample_rel(enc.embed(img.numpy()), g["emb_eval"])
        worst[os.path.basename(path)] = float(ps.max())
    assert len(worst) >= 8, worst
    # the 50-layer students (2048-wide sums, 2x2 final maps at 64 pixels) are held to 3e-3
    bad = {k: v for k, v in worst.items() if v > (3e-3 if ("r50" in k) else FP16_EMB_TOL)}
    assert not bad, (bad, worst)
