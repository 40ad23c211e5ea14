"""Pin the CPU oracle (oracle/vpd_oracle.py) against vectors produced by the
reference itself (oracle/gen_golden.py -> tests/golden/*.npz)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import vpd_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
# (c2_ / c3_ / c5_: the full-size cases -- minutes of CPU time each; test_fullsize_slice_* below checks a slice of them)
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "*r[0-9]*_c[0-9]_*.npz"))
               if not os.path.basename(p).startswith(("c2_", "c3_", "c5_")))


def sample_idx(numel, k=16):
    return np.unique(np.linspace(0, numel - 1, k).astype(np.int64))


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def build(meta):
    enc = O.procedural_state_dict(O.encoder_schema(meta["arch"], meta["c_in"], meta["emb_dim"]), meta["seed"])
    dec = O.procedural_state_dict(O.decoder_schema(meta["emb_dim"]), meta["seed"] + 7) if meta["motion"] else None
    img = O.synthetic_crops(meta["n"], meta["c_in"], meta["hw"], meta["seed"] + 1)
    tgt = O.synthetic_targets(meta["n"], meta["emb_dim"], meta["motion"], meta["seed"] + 2)
    return O.StudentOracle(meta["arch"], meta["c_in"], meta["emb_dim"], meta["motion"], enc, dec), img, tgt


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference(path):
    g = np.load(path)
    meta = json.loads(str(g["meta"]))
    torch.set_num_threads(4)

    # eval-mode embed() and eval epoch value
    st, img, tgt = build(meta)
    e = O.embed(st.enc, img.numpy(), meta["arch"], meta["c_in"] == 5 if meta["c_in"] in (3, 5) else meta["c_in"])
    assert e.dtype == np.float32 and e.shape == g["emb_eval"].shape
    assert rel_l2(e, g["emb_eval"]) < 1e-5
    assert abs(st.epoch([{"img": img, "emb": tgt}], train=False) - float(g["epoch_eval"])) < 1e-4 * abs(float(g["epoch_eval"]))

    # train-mode forward, loss, gradients, BN batch statistics
    st, img, tgt = build(meta)
    taps = {}
    loss, emb, out, grads = st.forward_loss(img, tgt, train=True, need_grad=True, taps=taps)
    assert rel_l2(emb.numpy(), g["emb_train"]) < 1e-5
    assert abs(loss - float(g["loss_train"])) < 1e-5 * abs(float(g["loss_train"]))
    for k in [k for k in g.files if k.startswith("bnmean/")]:
        name = k.split("/", 1)[1]
        assert rel_l2(taps[name][0].numpy(), g[k]) < 1e-4
        assert rel_l2(taps[name][1].numpy(), g["bnvar/" + name]) < 1e-4
    n_checked = 0
    for k in [k for k in g.files if k.startswith("gnorm/")]:
        name = k.split("/", 1)[1]
        gr = grads[name]
        ref_norm = float(g[k])
        assert abs(float(gr.double().norm()) - ref_norm) <= 2e-4 * max(ref_norm, 1e-6), name
        samp = gr.reshape(-1)[torch.from_numpy(sample_idx(gr.numel()))].numpy()
        assert np.allclose(samp, g["gsamp/" + name], rtol=2e-3, atol=2e-4 * max(ref_norm / np.sqrt(gr.numel()), 1e-8)), name
        n_checked += 1
    assert n_checked == len(st.params())

    # one full train step (epoch + AdamW + BN running stats), then two more
    st, img, tgt = build(meta)
    st.get_optimizer(meta["lr"])
    traj = [st.epoch([{"img": img, "emb": tgt}], train=True)]
    for k in [k for k in g.files if k.startswith("post/")]:
        name = k.split("/", 1)[1]
        if name.endswith("num_batches_tracked"):
            assert int(st.enc[name]) == int(g[k]) == 1
        else:
            assert rel_l2(st.enc[name].numpy(), g[k]) < 1e-5, name
    # AdamW's first step is sign-like: compare sampled weights with
    # an absolute tolerance of a fraction of lr rather than bitwise
    for k in [k for k in g.files if k.startswith("psamp/")]:
        name = k.split("/", 1)[1]
        p = st.params()[name]
        samp = p.reshape(-1)[torch.from_numpy(sample_idx(p.numel()))].numpy()
        assert np.allclose(samp, g[k], rtol=0, atol=0.6 * meta["lr"]), name
    traj += [st.epoch([{"img": img, "emb": tgt}], train=True) for _ in range(2)]
    assert np.allclose(traj, g["epoch_traj"], rtol=2e-2), (traj, g["epoch_traj"])
    assert abs(traj[0] - float(g["epoch_train"])) < 1e-5 * abs(traj[0])


def test_adamw_injected_grads():
    g = np.load(os.path.join(GOLDEN, "adamw_injected.npz"))
    p = {"w": torch.from_numpy(g["p0"].copy())}
    st = O.AdamWState(p)
    for t in range(3):
        O.adamw_update(p, {"w": torch.from_numpy(g["grads"][t].copy())}, st, float(g["lr"]))
        assert np.allclose(p["w"].numpy(), g["p_hist"][t], rtol=1e-6, atol=1e-7)
    assert np.allclose(st.m["w"].numpy(), g["m"], rtol=1e-6, atol=1e-9)
    assert np.allclose(st.v["w"].numpy(), g["v"], rtol=1e-6, atol=1e-12)
    d = json.loads(str(g["defaults"]))
    assert d["weight_decay"] == 0.01 and d["eps"] == 1e-8 and tuple(d["betas"]) == (0.9, 0.999)


def test_state_dict_schema_matches_reference():
    lines = open(os.path.join(GOLDEN, "format", "state_dict_schema.txt")).read().strip().split("\n")
    by_model = {}
    for ln in lines:
        model, key, shape, dtype = [s.strip() for s in ln.split("|")]
        by_model.setdefault(model, []).append((key, json.loads(shape), dtype))
    for model, (arch, c, d) in {"resnet34 c5 d128": ("resnet34", 5, 128), "resnet18 c3 d32": ("resnet18", 3, 32)}.items():
        sch = O.encoder_schema(arch, c, d)
        ours = [(k, list(s), "torch.int64" if kind == "bn_nbt" else "torch.float32") for k, (s, kind) in sch.items()]
        assert ours == by_model[model]
    ours = [(k, list(s), "torch.float32") for k, (s, _) in O.decoder_schema(128).items()]
    assert ours == by_model["fcnet d128"]
    assert len(O.encoder_schema("resnet34", 5, 128)) == 218


def test_reference_init_statistics():
    """Rows a3 / a4: the oracle's reference_init_state_dict against statistics of freshly constructed reference models
    (tests/golden/init_stats.json, written by oracle/gen_golden.py from models/rgb.py:8-43 + models/module.py:71-76).
    Random tensors are compared through what is deterministic about them: the std of a kaiming-normal sample of n
    elements (relative sampling error 1/sqrt(2n)), uniform bounds, the stem's identical slices; BN tensors exactly."""
    ref = json.load(open(os.path.join(GOLDEN, "init_stats.json")))
    for tag, (arch, c_in, D) in {"r34_c5_d128": ("resnet34", 5, 128), "r18_c3_d32": ("resnet18", 3, 32),
                                 "r50_c5_d32": ("resnet50", 5, 32)}.items():
        rows = ref[tag]
        sd = O.reference_init_state_dict(arch, c_in, D, 7)
        assert [k for k in rows if not k.startswith("__")] == list(sd.keys()) or \
            sorted(k for k in rows if not k.startswith("__")) == sorted(sd.keys())
        for k, v in sd.items():
            r = rows[k]
            assert list(v.shape) == r["shape"]
            kind = O.encoder_schema(arch, c_in, D)[k][1]
            if kind == "conv":
                n = v.numel() if not (k == "resnet.conv1.weight" and c_in != 3) else v.numel() // c_in
                tol = 5.0 / np.sqrt(2 * n)
                assert abs(float(v.double().std()) / r["std"] - 1) < 2 * tol + 0.02, (tag, k)
                assert abs(float(v.double().mean()) - r["mean"]) < 6 * r["std"] / np.sqrt(n), (tag, k)
            elif kind in ("bn_w", "bn_b", "bn_rm", "bn_rv", "bn_nbt"):
                assert float(v.double().min()) == r["min"] and float(v.double().max()) == r["max"], (tag, k)
            else:   # fc: U(-1/sqrt(in_features), +1/sqrt(in_features))
                b = rows["__fc__"]["bound"]
                assert abs(b - 1.0 / np.sqrt(512 * O.arch_expansion(arch))) < 1e-12
                assert r["min"] >= -b and r["max"] <= b and float(v.min()) >= -b and float(v.max()) <= b
                if v.numel() >= 4096:      # the bias has only D elements: bounds only
                    assert float(v.max()) > 0.9 * b and r["max"] > 0.9 * b
                    assert abs(float(v.double().std()) / (b / np.sqrt(3)) - 1) < 0.05 and abs(r["std"] / (b / np.sqrt(3)) - 1) < 0.05
        w = sd["resnet.conv1.weight"]
        assert rows["__stem__"]["c_in"] == c_in
        if c_in != 3:      # add_flow_to_model: every input-channel slice is the mean of the 3-channel kernel
            assert rows["__stem__"]["slices_identical"] and bool((w == w[:, :1]).all())
            # mean over 3 iid channels: std shrinks by sqrt(3)
            assert abs(rows["resnet.conv1.weight"]["std"] / (rows["__stem__"]["kaiming_std_3ch"] / np.sqrt(3)) - 1) < 0.05


def test_embed_contract():
    sd = O.procedural_state_dict(O.encoder_schema("resnet18", 5, 32), 1)
    x = O.synthetic_crops(1, 5, 64, 3)[0]
    assert O.embed(sd, x, "resnet18", True).shape == (1, 32)      # 3-D input promoted
    with pytest.raises(AssertionError):
        O.embed(sd, x[:3], "resnet18", True)
    with pytest.raises(AssertionError):
        O.embed(sd, x, "resnet18", False)


@pytest.mark.parametrize("name", ["c2_r34_c5_d128_m0_n256", "c3_r34_c6_d128_m1_n512", "c5_r34_c5_d128_n1000"])
def test_fullsize_slice_of_reference_embeddings(name):
    """The full-size goldens (256 / 512 / 1000 crops, generated by the reference: oracle/gen_golden.py FULLSIZE_CASES) are
    too big for the CPU suite as a whole; the eval forward is per crop, so the oracle on 12 crops of the batch must give
    those 12 rows (models/rgb.py:72-86)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    enc_sd = O.procedural_state_dict(O.encoder_schema(meta["arch"], meta["c_in"], meta["emb_dim"]), meta["seed"])
    img = O.synthetic_crops(meta["n"], meta["c_in"], meta["hw"], meta["seed"] + 1, O.FS_MEAN_STD if meta.get("norm") == "fs" else None)
    rows = np.asarray([0, 1, 2, 3, meta["n"] // 2, meta["n"] // 2 + 1, meta["n"] // 2 + 2, meta["n"] // 2 + 3,
                       meta["n"] - 4, meta["n"] - 3, meta["n"] - 2, meta["n"] - 1])
    e = O.embed(enc_sd, img[rows], meta["arch"], True if meta["c_in"] == 5 else meta["c_in"])
    err = np.abs(e - g["emb_eval"][rows]).max() / np.abs(g["emb_eval"][rows]).max()
    assert err <= 1e-5, err


def _wc_sample_idx(numel, k=512):
    return np.unique(np.linspace(0, numel - 1, k).astype(np.int64))      # oracle/gen_golden.py::sample_idx


@pytest.mark.parametrize("arch", ["resnet18", "resnet34", "resnet50"])
def test_oracle_gradients_match_the_reference_in_the_well_conditioned_regime(arch):
    """tests/golden/wc_grads_<arch>.npz: gradients of the reference's own modules (oracle/gen_golden.py::wellcond_case) with
    the last BatchNorm gamma of every residual branch scaled to 0.1, summed over three batches -- the regime in which the GPU
    tests gate the HIP backward.  The oracle must reproduce every sampled gradient element (models/module.py:35-130,
    train_vpd_model.py:87)."""
    g = np.load(os.path.join(GOLDEN, "wc_grads_%s.npz" % arch))
    sd = O.reference_init_state_dict(arch, 5, 32, 3)
    last = ".bn3.weight" if O.arch_expansion(arch) == 4 else ".bn2.weight"
    for k in sd:
        if k.endswith(last):
            sd[k] = sd[k] * 0.1
    acc, losses = {}, []
    for b in range(3):
        orc = O.StudentOracle(arch, 5, 32, False, {k: v.clone() for k, v in sd.items()}, None)
        img, tgt = O.synthetic_crops(8, 5, 128, 5 + 10 * b), O.synthetic_targets(8, 32, False, 6 + 10 * b)
        loss, _, _, grads = orc.forward_loss(img, tgt, train=True, need_grad=True)
        losses.append(float(loss))
        for k, v in grads.items():
            acc[k] = acc.get(k, 0.0) + v.double()
    assert np.allclose(losses, g["losses"], rtol=1e-5), (losses, g["losses"])
    keys = [k[len("gsamp/"):] for k in g.files if k.startswith("gsamp/")]
    assert len(keys) == sum(1 for k in acc if k.startswith("enc."))
    for k in keys:
        mine = acc["enc." + k].reshape(-1)
        samp = mine[torch.from_numpy(_wc_sample_idx(mine.numel()))].numpy()
        ref = g["gsamp/" + k].astype(np.float64)
        scale = float(g["gnorm/" + k]) / np.sqrt(mine.numel())          # RMS element of the tensor
        assert np.abs(samp - ref).max() <= 2e-3 * scale + 1e-7, (k, float(np.abs(samp - ref).max()), scale)
        assert abs(float(mine.norm()) - float(g["gnorm/" + k])) <= 2e-4 * float(g["gnorm/" + k]) + 1e-9, k


def test_fixture_recipe_reproduces_the_oracle_generators():
    """tests/golden/recipe.py (numpy only; what bench.py's parity block rebuilds a fixture's inputs with) == the oracle's
    procedural_state_dict / synthetic_crops, bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vpd_golden_recipe", os.path.join(GOLDEN, "recipe.py"))
    recipe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(recipe)
    for arch, c_in, d in (("resnet34", 5, 128), ("resnet18", 3, 32), ("resnet50", 6, 32)):
        sch = O.encoder_schema(arch, c_in, d)
        want = O.procedural_state_dict(sch, 11)
        got = recipe.procedural_weights({k: v[0] for k, v in sch.items()}, 11)
        assert list(got) == list(want)
        for k in want:
            assert np.array_equal(got[k], want[k].numpy()), k
    dsch = O.decoder_schema(32)
    want = O.procedural_state_dict(dsch, 5)
    got = recipe.procedural_weights({k: v[0] for k, v in dsch.items()}, 5)
    assert all(np.array_equal(got[k], want[k].numpy()) for k in want)
    for c_in in (3, 5, 6):
        assert np.array_equal(recipe.synthetic_crops(3, c_in, 32, 9), O.synthetic_crops(3, c_in, 32, 9).numpy())
