"""fp16 elements (libvpdhip_f16.so; RGBF_EmbeddingModel(dtype="fp16"), train_vpd_model.py / apply_vpd_model.py --dtype fp16): the
reference's own GPU precision -- fp16 autocast + GradScaler (train_vpd_model.py:79,105; models/util.py:55-57) -- through the C ABI.

bf16 keeps 8 significant bits, fp16 11: against the reference's fp32 CPU numbers the gates of this file are HARD per-tensor gates where
the bf16 gates of tests/test_model_gpu.py have to be statistical (parity soft spot 1).  Training runs through the
LossScaler exactly as the reference's step() runs its GradScaler: scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update().
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import vpd_oracle as O
from tests.test_model_gpu import CASES, GOLDEN, _dump, _group_of, _wc_sample_idx, cosine, per_sample_rel, rel_l2

pytestmark = pytest.mark.gpu

# measured (profiles/r06_fp16_parity.txt): eval embeddings 4-9e-4 per sample; train loss 2e-6..3e-4; running statistics 4e-4..1.6e-3
# (50-layer: 1e-2); whole-gradient error against the fp32 reference 0.11-0.27 (bf16: 0.34-0.71) with cos >= 0.93 per tensor (bf16: 0.57)
FP16_EMB_TOL = 1.5e-3
FP16_LOSS_TOL = 1e-3
DEEP = ("resnet50", "resnet101", "wide_resnet50_2", "wide_resnet101_2")


def build_fp16(meta):
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.trainer import ModelTrainer
    enc_sd = O.procedural_state_dict(O.encoder_schema(meta["arch"], meta["c_in"], meta["emb_dim"]), meta["seed"])
    dec_sd = O.procedural_state_dict(O.decoder_schema(meta["emb_dim"]), meta["seed"] + 7) if meta["motion"] else None
    img = O.synthetic_crops(meta["n"], meta["c_in"], meta["hw"], meta["seed"] + 1,
                            O.FS_MEAN_STD if meta.get("norm") == "fs" else None)
    tgt = O.synthetic_targets(meta["n"], meta["emb_dim"], meta["motion"], meta["seed"] + 2)
    enc = RGBF_EmbeddingModel(meta["arch"], meta["emb_dim"], meta["c_in"] != 3, "cuda",
                              in_channels=None if meta["c_in"] in (3, 5) else meta["c_in"], dtype="fp16")
    enc.load_state_dict(enc_sd)
    tr = ModelTrainer(enc, meta["motion"])
    if meta["motion"]:
        tr.fcn_time.load_state_dict(dec_sd)
    return enc, tr, enc_sd, dec_sd, img, tgt


def scaled_backward(tr, scaler, loss):
    """scaler.scale(loss).backward() as the reference's step() runs it, then the true gradients: .grad / scale (what
    scaler.step() hands a torch optimizer).  Returns {name: fp64 cpu tensor}."""
    scaler.scale(loss).backward()
    torch.cuda.synchronize()
    eng = tr.encoder.engine
    s = scaler.get_scale()
    assert eng.loss_scale == s
    out = {"enc." + n: q.grad.detach().cpu().double() / s for n, q in tr.encoder.named_parameters()}
    if hasattr(tr, "fcn_time"):
        out.update({"dec." + n: q.grad.detach().cpu().double() / s for n, q in tr.fcn_time.named_parameters()})
    eng.loss_scale = 1.0
    return out


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_fp16_student_matches_the_reference(path):
    from vpd_amd.models.util import LossScaler
    g = np.load(path)
    meta = json.loads(str(g["meta"]))
    deep = meta["arch"] in DEEP
    rec = {"meta": meta}
    enc, tr, enc_sd, dec_sd, img, tgt = build_fp16(meta)
    # eval-mode embed() and the eval epoch value
    e = enc.embed(img.numpy())
    ps = per_sample_rel(e, g["emb_eval"])
    rec["emb_eval_per_sample_max"] = float(ps.max())
    assert ps.max() <= FP16_EMB_TOL, float(ps.max())
    ev = tr.epoch([{"img": img, "emb": tgt}])
    assert abs(ev - float(g["epoch_eval"])) <= FP16_LOSS_TOL * abs(float(g["epoch_eval"]))
    # train-mode forward, loss, scaled backward
    enc, tr, enc_sd, dec_sd, img, tgt = build_fp16(meta)
    optimizer, scaler = tr.get_optimizer(meta["lr"])
    assert isinstance(scaler, LossScaler) and scaler.get_scale() == 256.0
    enc.train()
    loss = tr._forward_loss(img, tgt, train=True)
    l_hip = loss.item()
    rec["loss_train"] = [l_hip, float(g["loss_train"])]
    assert abs(l_hip - float(g["loss_train"])) <= FP16_LOSS_TOL * abs(float(g["loss_train"])), rec["loss_train"]
    grads = scaled_backward(tr, scaler, loss)
    orc = O.StudentOracle(meta["arch"], meta["c_in"], meta["emb_dim"], meta["motion"], enc_sd, dec_sd)
    _, _, _, grads_ref = orc.forward_loss(img, tgt, train=True, need_grad=True)
    err, flat_h, flat_r = {}, [], []
    for name, gref in grads_ref.items():
        got = grads[name].numpy()
        err[name] = [round(rel_l2(got, gref.numpy()), 4), round(cosine(got, gref.numpy()), 4)]
        flat_h.append(got.ravel()); flat_r.append(gref.numpy().ravel())
    ge = np.asarray(list(err.values()))
    rec["grad_err_cos"] = err
    rec["grad_flat_err"] = rel_l2(np.concatenate(flat_h), np.concatenate(flat_r))
    rec["grad_err_max"], rec["grad_cos_min"] = float(ge[:, 0].max()), float(ge[:, 1].min())
    # running statistics after one train-mode forward vs the reference after one step
    sd = enc.state_dict()
    worst = 0.0
    for k in [k for k in g.files if k.startswith("post/") and not k.endswith("num_batches_tracked")]:
        worst = max(worst, rel_l2(sd[k.split("/", 1)[1]].cpu().numpy(), g[k]))
    rec["running_stats_rel_l2_max"] = worst
    # three steps through ModelTrainer.epoch / step() with the scaler, as train_vpd_model.py runs them
    enc, tr, enc_sd, dec_sd, img, tgt = build_fp16(meta)
    optimizer, scaler = tr.get_optimizer(meta["lr"])
    traj = [tr.epoch([{"img": img, "emb": tgt}], optimizer=optimizer, scaler=scaler) for _ in range(3)]
    rec["epoch_traj"] = [traj, g["epoch_traj"].tolist()]
    _dump("fp16_" + meta["name"], rec)
    # HARD gates (per tensor, no emulation to hide behind).  The early layers of an untrained student on 5-8 crops amplify any
    # forward rounding: 0.45 / cos 0.88 there; the 50-layer students sit in the chaotic regime (fp32 vs fp32 on
    # another host's convolutions already differs by 1e-2 in the gradient norms): loss and statistics only
    assert worst <= (2.5e-2 if deep else 4e-3), worst
    if not deep:
        assert rec["grad_flat_err"] <= 0.35 and rec["grad_err_max"] <= 0.45 and rec["grad_cos_min"] >= 0.88, \
            (rec["grad_flat_err"], rec["grad_err_max"], rec["grad_cos_min"])
    assert abs(traj[0] - float(g["epoch_traj"][0])) <= FP16_LOSS_TOL * abs(traj[0])
    # second step: measured against how far the reference's own loss moved in that step (the 8-crop 64-pixel ResNet-34 case drops from
    # 44.9 to 15.0 in one step: any rounding of the first update is amplified) -- 2.5 % of the step (50-layer: 5 %) + 1e-3 of the value
    ref0, ref1 = float(g["epoch_traj"][0]), float(g["epoch_traj"][1])
    assert abs(traj[1] - ref1) <= (0.05 if deep else 0.025) * abs(ref0 - ref1) + 1e-3 * abs(ref1), rec["epoch_traj"]


@pytest.mark.parametrize("arch", ["resnet18", "resnet34", "resnet50"])
def test_fp16_backward_matches_the_reference_gradients(arch):
    """The reference's OWN sampled gradients in the well-conditioned regime (tests/golden/wc_grads_<arch>.npz; see
    test_backward_matches_the_reference_gradients for the recipe): per network stage cosine and projection of the fp16 path, run
    through the LossScaler.  bf16 is held to cos >= 0.95, 1 +- 6 %."""
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.trainer import ModelTrainer
    g = np.load(os.path.join(GOLDEN, "wc_grads_%s.npz" % arch))
    sd = O.reference_init_state_dict(arch, 5, 32, 3)
    last = ".bn3.weight" if O.arch_expansion(arch) == 4 else ".bn2.weight"
    for k in sd:
        if k.endswith(last):
            sd[k] = sd[k] * 0.1
    enc = RGBF_EmbeddingModel(arch, 32, True, "cuda", dtype="fp16")
    tr = ModelTrainer(enc, False)
    _, scaler = tr.get_optimizer(5e-4)
    acc, losses = {}, []
    for b in range(3):
        enc.load_state_dict(sd)
        enc.train()
        img, tgt = O.synthetic_crops(8, 5, 128, 5 + 10 * b), O.synthetic_targets(8, 32, False, 6 + 10 * b)
        loss = tr._forward_loss(img, tgt, train=True)
        losses.append(loss.item())
        for n, q in scaled_backward(tr, scaler, loss).items():
            acc[n[4:]] = acc.get(n[4:], 0.0) + q
    for lh, lr in zip(losses, g["losses"]):
        assert abs(lh - lr) <= 1e-3 * lr, (losses, g["losses"])
    stage = {}
    for n in acc:
        ref = g["gsamp/" + n].astype(np.float64)
        mine = acc[n].reshape(-1)[torch.from_numpy(_wc_sample_idx(acc[n].numel()))].numpy()
        v = stage.setdefault(_group_of(n), [0.0, 0.0, 0.0])
        v[0] += float((mine * ref).sum()); v[1] += float((ref * ref).sum()); v[2] += float((mine * mine).sum())
    res = {k: (round(v[0] / (v[1] * v[2]) ** 0.5, 4), round(v[0] / v[1], 4)) for k, v in stage.items()}
    _dump("fp16_reference_grads_%s" % arch, {"arch": arch, "columns": ["cos", "projection"], "hip_vs_reference": res})
    assert all(c >= 0.99 and abs(pj - 1) <= 0.02 for c, pj in res.values()), res


def test_loss_scaler_semantics():
    """(i) scale(loss).backward() leaves every .grad = scale x the unscaled backward's -- a power-of-two scale commutes with every
    rounding of the pass, so wherever nothing underflowed without it the two agree to the bit; (ii) scaler.step() with a torch
    optimizer (un-scaled p.grad) and with the fused AdamW (un-scaled in the kernel) move the parameters alike; (iii) step() takes
    scaler=None or a LossScaler, nothing else."""
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    from vpd_amd.models.util import LossScaler, step
    from vpd_amd.trainer import ModelTrainer
    sd = O.procedural_state_dict(O.encoder_schema("resnet18", 5, 32), 3)
    img, tgt = O.synthetic_crops(8, 5, 64, 4), O.synthetic_targets(8, 32, False, 5)

    def fresh():
        enc = RGBF_EmbeddingModel("resnet18", 32, True, "cuda", dtype="fp16")
        enc.load_state_dict(sd)
        enc.train()
        return enc, ModelTrainer(enc, False)

    enc, tr = fresh()
    tr._forward_loss(img, tgt, train=True).backward()
    torch.cuda.synchronize()
    plain = {n: q.grad.detach().clone() for n, q in enc.named_parameters()}
    enc, tr = fresh()
    sc = LossScaler(enc.engine, 1024.0)
    sc.scale(tr._forward_loss(img, tgt, train=True)).backward()
    torch.cuda.synchronize()
    same, total = 0, 0
    for n, q in enc.named_parameters():
        a, b = q.grad.detach() / 1024.0, plain[n]
        same += int((a == b).sum()); total += a.numel()
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 2e-2, n
    # (measured: 84 % of the 11.2 M gradient elements agree to the bit; the rest are where the UNSCALED pass went through fp16
    #  subnormals -- what the scale is for)
    assert same >= 0.75 * total, (same, total)
    enc.engine.loss_scale = 1.0
    # (ii) fused AdamW vs torch.optim.AdamW behind the same scaler
    enc_a, tr_a = fresh()
    opt_a, sc_a = tr_a.get_optimizer(1e-3)
    step(opt_a, sc_a, tr_a._forward_loss(img, tgt, train=True))
    enc_b, tr_b = fresh()
    opt_b, sc_b = torch.optim.AdamW(list(enc_b.parameters()), lr=1e-3), LossScaler(enc_b.engine)
    step(opt_b, sc_b, tr_b._forward_loss(img, tgt, train=True))
    torch.cuda.synchronize()
    for (n, qa), (_, qb) in zip(enc_a.named_parameters(), enc_b.named_parameters()):
        assert torch.allclose(qa, qb, rtol=0, atol=2e-6), n
    moved = max(float((q.detach() - sd[n].cuda()).abs().max()) for n, q in enc_a.named_parameters())
    assert moved > 5e-4
    # (iii)
    with pytest.raises(ValueError):
        step(opt_a, torch.amp.GradScaler("cuda", enabled=False), tr_a._forward_loss(img, tgt, train=True))


def test_fp16_full_size_train_step_matches_the_reference():
    """BASELINE configs[1] at its full 256 crops in fp16: train-mode embeddings, loss, gradient norms per stage, running statistics
    against what the reference produced (tests/golden/c2_r34_c5_d128_m0_n256.npz).  bf16: 6e-2 / 1e-2 / +-15 % / 2e-2."""
    g = np.load(os.path.join(GOLDEN, "c2_r34_c5_d128_m0_n256.npz"))
    meta = json.loads(str(g["meta"]))
    enc, tr, _, _, img, tgt = build_fp16(meta)
    _, scaler = tr.get_optimizer(5e-4)
    enc.train()
    eng = enc.engine
    emb = eng.forward_train(img.cuda(), tgt.cuda(), motion=meta["motion"], accumulate_loss=False).clone()
    l_hip = float(eng.loss_step.item())
    ps = per_sample_rel(emb.cpu().numpy(), g["emb_train"])
    rec = {"loss_train": [l_hip, float(g["loss_train"])], "emb_train_per_sample_max": float(ps.max())}
    eng.loss_scale = scaler.get_scale()
    eng.backward()
    torch.cuda.synchronize()
    stage = lambda k: k.split(".")[2] if k.startswith("enc.resnet.layer") else ("stem" if "conv1" in k or "bn1" in k else "fc")
    hip2, ref2 = {}, {}
    for k in [f[len("gnorm/"):] for f in g.files if f.startswith("gnorm/")]:
        got = enc.get_parameter(k[4:]).grad / scaler.get_scale()
        s_ = stage(k)
        hip2[s_] = hip2.get(s_, 0.0) + float(got.double().pow(2).sum())
        ref2[s_] = ref2.get(s_, 0.0) + float(g["gnorm/" + k]) ** 2
    rec["grad_norm_ratio_by_stage"] = {s_: (hip2[s_] / ref2[s_]) ** 0.5 for s_ in ref2}
    sd = enc.state_dict()
    rec["running_stats_rel_l2_max"] = max(rel_l2(sd[f[len("post/"):]].cpu().numpy(), g[f]) for f in g.files
                                          if f.startswith("post/") and not f.endswith("num_batches_tracked"))
    _dump("fp16_fullsize_c2", rec)
    assert abs(l_hip - float(g["loss_train"])) <= FP16_LOSS_TOL * abs(float(g["loss_train"])), rec["loss_train"]
    assert ps.max() <= 1e-2, float(ps.max())
    assert all(0.95 <= v <= 1.05 for v in rec["grad_norm_ratio_by_stage"].values()), rec["grad_norm_ratio_by_stage"]
    assert rec["running_stats_rel_l2_max"] <= 4e-3, rec["running_stats_rel_l2_max"]
