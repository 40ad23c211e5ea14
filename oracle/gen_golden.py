#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (authoring container only).

This is the only file in the repository that touches ``/root/reference``.  It
imports the reference's own ``RGBF_EmbeddingModel``, ``ModelTrainer``, ``FCNet``
and ``step`` and drives them on seeded inputs; the outputs are committed as
small fixtures so that they travel to the GPU box, where the reference does
not exist.  Three third-party imports of the reference are absent from this
image and are stubbed before import:

* ``cv2``, ``efficientnet_pytorch`` -- never reached on synthetic batches;
* ``torchvision`` -- ``transforms`` (unused on synthetic batches) and
  ``models.resnet.{BasicBlock, Bottleneck, conv1x1}``.  The topology and
  initialisation come from the reference's OWN in-repo ``models.module.ResNet``
  (models/module.py:35-130); only the residual block (conv3x3-BN-ReLU-conv3x3-
  BN-add-ReLU, expansion 1) is supplied here, written from its published
  definition.

Usage:  python oracle/gen_golden.py          (writes tests/golden/)
"""
import json
import os
import pickle
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import vpd_oracle as O  # noqa: E402  (schemas / procedural weights / synthetic inputs)


# ---------------------------------------------------------------------------
# stubs for the three absent third-party modules
# ---------------------------------------------------------------------------
def _install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    sys.modules["cv2"] = cv2

    eff = types.ModuleType("efficientnet_pytorch")
    eff.EfficientNet = type("EfficientNet", (), {})
    eff.model = types.ModuleType("efficientnet_pytorch.model")
    sys.modules["efficientnet_pytorch"] = eff
    sys.modules["efficientnet_pytorch.model"] = eff.model

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    for name in ("Normalize", "Compose", "ColorJitter", "RandomResizedCrop"):
        setattr(tvt, name, type(name, (), {"__init__": lambda self, *a, **k: None}))
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")

    def conv1x1(i, o, stride=1):
        return nn.Conv2d(i, o, kernel_size=1, stride=stride, bias=False)

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1,
                     base_width=64, dilation=1, norm_layer=None):
            super().__init__()
            norm_layer = norm_layer or nn.BatchNorm2d
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = norm_layer(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = norm_layer(planes)
            self.downsample = downsample
            self.stride = stride

        def forward(self, x):
            idn = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return self.relu(out + idn)

    class Bottleneck(nn.Module):
        """torchvision's Bottleneck from its published definition ("ResNet v1.5": the 3x3 conv carries the stride):
        conv1x1 -> BN -> ReLU -> conv3x3(stride) -> BN -> ReLU -> conv1x1(x4) -> BN -> (+identity) -> ReLU."""
        expansion = 4

        def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1,
                     base_width=64, dilation=1, norm_layer=None):
            super().__init__()
            norm_layer = norm_layer or nn.BatchNorm2d
            width = int(planes * (base_width / 64.0)) * groups
            self.conv1 = conv1x1(inplanes, width)
            self.bn1 = norm_layer(width)
            self.conv2 = nn.Conv2d(width, width, 3, stride, dilation, groups=groups, bias=False, dilation=dilation)
            self.bn2 = norm_layer(width)
            self.conv3 = conv1x1(width, planes * self.expansion)
            self.bn3 = norm_layer(planes * self.expansion)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = downsample
            self.stride = stride

        def forward(self, x):
            idn = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.relu(self.bn2(self.conv2(out)))
            out = self.bn3(self.conv3(out))
            return self.relu(out + idn)

    tvr.BasicBlock, tvr.Bottleneck, tvr.conv1x1 = BasicBlock, Bottleneck, conv1x1
    for name in ("resnet18", "resnet34", "resnet50", "resnet101", "wide_resnet50_2", "wide_resnet101_2"):
        setattr(tvm, name, None)
    tvm.resnet = tvr
    tv.transforms, tv.models = tvt, tvm
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.models": tvm, "torchvision.models.resnet": tvr})
    return BasicBlock, Bottleneck


def _import_reference():
    basic_block, bottleneck = _install_stubs()
    sys.path.insert(0, REF)
    import models.module as ref_module
    import models.rgb as ref_rgb
    import models.util as ref_util
    import train_vpd_model as ref_train
    import util.io as ref_io
    import action_dataset.load as ref_load

    for arch, layers in (("resnet18", [2, 2, 2, 2]), ("resnet34", [3, 4, 6, 3])):
        cfg = ref_module.ENCODER_ARCH[arch]
        ref_module.ENCODER_ARCH[arch] = cfg._replace(
            pretrained_init=(lambda L: (lambda pretrained=False: ref_module.ResNet(basic_block, L, 3, 1000)))(layers))
    # Bottleneck archs: the reference's own ResNet with its own ENCODER_ARCH layer lists / widths
    for arch in ("resnet50", "resnet101", "wide_resnet50_2", "wide_resnet101_2"):
        cfg = ref_module.ENCODER_ARCH[arch]
        ref_module.ENCODER_ARCH[arch] = cfg._replace(
            pretrained_init=(lambda c: (lambda pretrained=False: ref_module.ResNet(
                bottleneck, c.layers, 3, 1000, width_per_group=c.width_per_group)))(cfg))
    return ref_module, ref_rgb, ref_util, ref_train, ref_io, ref_load


# ---------------------------------------------------------------------------
def sample_idx(numel: int, k: int = 16) -> np.ndarray:
    return np.unique(np.linspace(0, numel - 1, k).astype(np.int64))


def _samples(t: torch.Tensor) -> np.ndarray:
    flat = t.detach().reshape(-1)
    return flat[torch.from_numpy(sample_idx(flat.numel()))].numpy().astype(np.float32)


CASES = [
    # name, arch, c_in, D, motion, N, HW, lr
    ("r34_c5_d128_m1_n8", "resnet34", 5, 128, True, 8, 128, 5e-4),
    ("r34_c5_d128_m0_n5", "resnet34", 5, 128, False, 5, 128, 5e-4),
    ("r34_c3_d32_m0_n8_hw64", "resnet34", 3, 32, False, 8, 64, 5e-4),
    ("r18_c5_d32_m1_n8", "resnet18", 5, 32, True, 8, 128, 5e-4),
    ("r18_c3_d128_m0_n5_hw64", "resnet18", 3, 128, False, 5, 64, 1e-3),
    ("r18_c5_d128_m1_n6_hw64", "resnet18", 5, 128, True, 6, 64, 5e-4),
    # Bottleneck students (row f2)
    ("r50_c5_d32_m0_n8", "resnet50", 5, 32, False, 8, 128, 5e-4),
    ("wr50_c3_d32_m1_n6", "wide_resnet50_2", 3, 32, True, 6, 128, 5e-4),
    # BASELINE configs[2]: --motion two-stream, 6-channel input (SURVEY 8d runs C=6 in addition to C=5)
    ("r34_c6_d128_m1_n6", "resnet34", 6, 128, True, 6, 128, 5e-4),
    # BASELINE configs[0] at its full size: 64 crops of 5x128x128, 128-d teacher embeddings (+ motion: 256-d targets), one
    # (here: three) train step(s) of the reference on the CPU
    ("c1_r34_c5_d128_m1_n64", "resnet34", 5, 128, True, 64, 128, 5e-4),
]
# Full-size cases: BASELINE configs[1]-[4] at their own per-GPU sizes, so that the HIP path is compared with
# values the REFERENCE produced at 256 / 512 / 1000 crops, not only with size-independent properties.
#   level "full":  everything run_case stores except the post-step eval embeddings and the loss trajectory
#   level "light": eval embeddings + train-mode embeddings, loss and BatchNorm taps (no backward: halves the CPU time / memory)
#   level "eval":  eval embeddings only (the apply path)
# name, arch, c_in, D, motion, N, HW, lr, level, crop normalisation
FULLSIZE_CASES = [
    ("c2_r34_c5_d128_m0_n256", "resnet34", 5, 128, False, 256, 128, 5e-4, "full", None),
    ("c3_r34_c5_d128_m1_n512_fs", "resnet34", 5, 128, True, 512, 128, 5e-4, "light", "fs"),
    ("c3_r34_c6_d128_m1_n512", "resnet34", 6, 128, True, 512, 128, 5e-4, "light", None),
    ("c5_r34_c5_d128_n1000", "resnet34", 5, 128, False, 1000, 128, 5e-4, "eval", None),
]
TAP_BNS = ["resnet.bn1", "resnet.layer1.0.bn1", "resnet.layer2.0.downsample.1",
           "resnet.layer3.1.bn2", "resnet.layer4.1.bn2"]


def run_case(ref, name, arch, c_in, D, motion, N, HW, lr, seed, level="all", norm=None):
    ref_module, ref_rgb, ref_util, ref_train, _, _ = ref
    enc_sd = O.procedural_state_dict(O.encoder_schema(arch, c_in, D), seed)
    dec_sd = O.procedural_state_dict(O.decoder_schema(D), seed + 7) if motion else None
    img = O.synthetic_crops(N, c_in, HW, seed + 1, O.FS_MEAN_STD if norm == "fs" else None)
    tgt = O.synthetic_targets(N, D, motion, seed + 2)

    def build():
        if c_in in (3, 5):
            enc = ref_rgb.RGBF_EmbeddingModel(arch, D, c_in == 5, "cpu")
        else:
            # the reference hard-codes 5 input channels (models/rgb.py:21, :25): build its 3-channel student and swap the
            # stem by add_flow_to_model's own recipe with c_in in place of 5 -- channel mean of the 3-channel kernel
            # expanded (the weights are overwritten by the procedural state_dict anyway; the modules that run are the
            # reference's).  embed()'s channel assert (models/rgb.py:79-82) only knows 3 / 5: use_flow=True + patched check.
            enc = ref_rgb.RGBF_EmbeddingModel(arch, D, False, "cpu")
            old = enc.resnet.conv1
            new = nn.Conv2d(c_in, old.out_channels, old.kernel_size, old.stride, old.padding, bias=False)
            new.weight.data = old.weight.data.mean(dim=1, keepdim=True).expand(
                old.weight.shape[:1] + (c_in,) + old.weight.shape[2:]).contiguous()
            enc.resnet.conv1 = new

            def embed_any(x, enc=enc):
                enc.eval()
                with torch.no_grad():
                    return enc(torch.as_tensor(x, dtype=torch.float32)).cpu().numpy()
            enc.embed = embed_any
        enc.load_state_dict(enc_sd)
        tr = ref_train.ModelTrainer(enc, motion)
        if motion:
            tr.fcn_time.load_state_dict(dec_sd)
        return enc, tr

    out = {"meta": json.dumps(dict(name=name, arch=arch, c_in=c_in, emb_dim=D, motion=motion,
                                   n=N, hw=HW, lr=lr, seed=seed, level=level, norm=norm))}

    # (1) eval-mode embeddings through the reference's embed()  (models/rgb.py:72-86)
    enc, tr = build()
    out["emb_eval"] = enc.embed(img.numpy())
    if level == "eval":
        return out
    # eval-mode epoch value (optimizer=None): train_vpd_model.py:70-76
    out["epoch_eval"] = np.float64(tr.epoch([{"img": img, "emb": tgt}]))

    # (2) train-mode forward + loss + backward with the reference's modules
    enc, tr = build()
    enc.train()
    taps = {}
    hooks = []
    mods = dict(enc.named_modules())
    for bn_name in TAP_BNS:
        if bn_name in mods:
            def hk(m, inp, outp, key=bn_name):
                x = inp[0].detach()
                taps[key] = (x.mean(dim=(0, 2, 3)).numpy(), x.var(dim=(0, 2, 3), unbiased=False).numpy())
            hooks.append(mods[bn_name].register_forward_hook(hk))
    if motion:
        tr.fcn_time.train()
    if level == "light":
        with torch.no_grad():
            emb = enc(img)
            pred = tr.fcn_time(emb) if motion else emb
            loss = torch.nn.functional.mse_loss(pred, tgt, reduction="sum")
    else:
        emb = enc(img)
        pred = tr.fcn_time(emb) if motion else emb
        loss = torch.nn.functional.mse_loss(pred, tgt, reduction="sum")
        loss.backward()
    for h in hooks:
        h.remove()
    out["emb_train"] = emb.detach().numpy()
    out["loss_train"] = np.float64(loss.item())
    for k, (m, v) in taps.items():
        out["bnmean/" + k] = m
        out["bnvar/" + k] = v
    if level == "light":
        return out
    named = [("enc." + k, p) for k, p in enc.named_parameters()]
    if motion:
        named += [("dec." + k, p) for k, p in tr.fcn_time.named_parameters()]
    for k, p in named:
        out["gnorm/" + k] = np.float64(p.grad.double().norm().item())
        out["gsamp/" + k] = _samples(p.grad)

    # (3) one full reference train step: ModelTrainer.epoch + get_optimizer + step
    enc, tr = build()
    optimizer, scaler = tr.get_optimizer(lr)
    assert scaler is None
    out["epoch_train"] = np.float64(tr.epoch([{"img": img, "emb": tgt}], optimizer=optimizer, scaler=scaler))
    for k, v in enc.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["post/" + k] = v.numpy().copy()
        elif k.endswith("num_batches_tracked"):
            out["post/" + k] = v.numpy().copy()
    named = [("enc." + k, p) for k, p in enc.named_parameters()]
    if motion:
        named += [("dec." + k, p) for k, p in tr.fcn_time.named_parameters()]
    for k, p in named:
        out["psamp/" + k] = _samples(p)
    if level == "full":
        return out
    out["emb_eval_post"] = enc.embed(img.numpy())
    # second + third step losses (three-step trajectory, SURVEY 8c probe)
    traj = [float(out["epoch_train"])]
    for _ in range(2):
        traj.append(float(tr.epoch([{"img": img, "emb": tgt}], optimizer=optimizer, scaler=scaler)))
    out["epoch_traj"] = np.asarray(traj, np.float64)
    return out


def init_case(ref):
    """Rows a3 / a4: statistics of FRESHLY CONSTRUCTED reference models (models/rgb.py:8-43 add_flow_to_model /
    replace_last_layer on top of models/module.py:71-76) and of the motion head FCNet (models/module.py:133-156), so that
    the product's reset_parameters() is pinned to what the reference's constructors produce."""
    ref_module, ref_rgb, _, _, _, _ = ref
    res = {}
    for tag, arch, c_in, D in (("r34_c5_d128", "resnet34", 5, 128), ("r18_c3_d32", "resnet18", 3, 32),
                               ("r50_c5_d32", "resnet50", 5, 32)):
        torch.manual_seed(1234)
        m = ref_rgb.RGBF_EmbeddingModel(arch, D, c_in == 5, "cpu")
        rows = {}
        for k, v in m.state_dict().items():
            v64 = v.double()
            rows[k] = dict(shape=list(v.shape), dtype=str(v.dtype), mean=float(v64.mean()), std=float(v64.std()) if v.numel() > 1 else 0.0,
                           min=float(v64.min()), max=float(v64.max()))
        w = m.state_dict()["resnet.conv1.weight"]
        rows["__stem__"] = dict(slices_identical=bool((w == w[:, :1]).all()), c_in=int(w.shape[1]),
                                kaiming_std_3ch=float(np.sqrt(2.0 / (64 * 49))))
        fc = m.state_dict()["resnet.fc.weight"]
        rows["__fc__"] = dict(in_features=int(fc.shape[1]), bound=float(1.0 / np.sqrt(fc.shape[1])))
        res[tag] = rows
    torch.manual_seed(4321)
    fcn = ref_module.FCNet(128, [128, 128], 256, dropout=0)
    res["fcnet_d128"] = {k: dict(shape=list(v.shape), mean=float(v.double().mean()), std=float(v.double().std()),
                                 min=float(v.min()), max=float(v.max())) for k, v in fcn.state_dict().items()}
    return res


def adamw_case(ref):
    """AdamW with injected grads for t=1..3 through torch.optim.AdamW exactly
    as get_optimizer builds it (train_vpd_model.py:100-105)."""
    rs = np.random.RandomState(11)
    p0 = rs.standard_normal(4096).astype(np.float32)
    gs = (rs.standard_normal((3, 4096)) * np.array([1.0, 1e-3, 10.0])[:, None]).astype(np.float32)
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([p], lr=5e-4)
    hist = []
    for t in range(3):
        p.grad = torch.from_numpy(gs[t].copy())
        opt.step()
        opt.zero_grad()
        hist.append(p.detach().numpy().copy())
    st = opt.state[p]
    return dict(p0=p0, grads=gs, p_hist=np.stack(hist), m=st["exp_avg"].numpy(), v=st["exp_avg_sq"].numpy(),
                lr=np.float64(5e-4), defaults=json.dumps({k: v for k, v in opt.defaults.items()
                                                          if isinstance(v, (int, float, bool, tuple, list))}))


def format_case(ref):
    """Reference apply-loop body (apply_vpd_model.py:152-178) on a stub dataset
    -> pickles via the reference's store_pickle, re-read via group_by_frame;
    plus config.json / loss.json via the reference's store_json."""
    ref_module, ref_rgb, ref_util, ref_train, ref_io, ref_load = ref
    arch, D, c_in = "resnet18", 32, 5
    enc_sd = O.procedural_state_dict(O.encoder_schema(arch, c_in, D), 5)
    enc = ref_rgb.RGBF_EmbeddingModel(arch, D, True, "cpu")
    enc.load_state_dict(enc_sd)
    videos = ["vidA", "vidB", "vidC"]
    frames = {0: [3, 1, 2, 0], 1: [10, 12, 11, 14], 2: [5, 6, 8, 7]}
    fdir = os.path.join(OUT, "format")
    os.makedirs(fdir, exist_ok=True)
    res = {}
    for k, tag in ((2, "k2"), (1, "k1")):
        tasks = [(vid, fr) for vid in range(3) for fr in frames[vid]]
        imgs = O.synthetic_crops(len(tasks) * k, c_in, 64, 21).reshape(len(tasks), k, c_in, 64, 64)
        all_embs = [list() for _ in videos]
        bs = 5
        for s in range(0, len(tasks), bs):
            batch_img = imgs[s:s + bs]
            n_batch, kk, w, h, d = batch_img.shape
            batch_embs = enc.embed(batch_img.view(-1, w, h, d)).reshape((n_batch, kk, -1))
            for i in range(n_batch):
                vid, fr = tasks[s + i]
                all_embs[vid].append((fr, batch_embs[i, :, :] if kk > 1 else batch_embs[i, 0, :], {}))
        for video_name, embs in zip(videos, all_embs):
            embs.sort()
            path = os.path.join(fdir, "%s.%s.emb.pkl" % (video_name, tag))
            ref_io.store_pickle(path, embs)
            dense, mask = ref_load.group_by_frame(ref_io.load_pickle(path))
            res["dense/%s/%s" % (tag, video_name)] = dense
            res["mask/%s/%s" % (tag, video_name)] = mask
    res["tasks"] = np.asarray([(vid, fr) for vid in range(3) for fr in frames[vid]], np.int64)
    ref_io.store_json(os.path.join(fdir, "config.json"), {
        "num_epochs": 2, "batch_size": 8, "learning_rate": 5e-4, "img_dim": 64, "use_flow": True,
        "motion": False, "emb_dim": D, "encoder_arch": arch, "rgb_mean_std": O.DIVING48_MEAN_STD})
    ref_io.store_json(os.path.join(fdir, "loss.json"), [
        {"epoch": 1, "train": 1.5, "val": 2.5, "dataset_train": [("diving48", 1.5)],
         "dataset_val": [("diving48", 2.5)]}])
    # state_dict key/shape/dtype list of the reference's models (text fixture)
    lines = []
    for arch_, c_, d_ in (("resnet34", 5, 128), ("resnet18", 3, 32)):
        m = ref_rgb.RGBF_EmbeddingModel(arch_, d_, c_ == 5, "cpu")
        for k_, v_ in m.state_dict().items():
            lines.append("%s c%d d%d | %s | %s | %s" % (arch_, c_, d_, k_, list(v_.shape), str(v_.dtype)))
    fc = ref_module.FCNet(128, [128, 128], 256, dropout=0)
    for k_, v_ in fc.state_dict().items():
        lines.append("fcnet d128 | %s | %s | %s" % (k_, list(v_.shape), str(v_.dtype)))
    with open(os.path.join(fdir, "state_dict_schema.txt"), "w") as fp:
        fp.write("\n".join(lines) + "\n")
    return res


def reader_case(ref):
    """Row f4's reader: pickles with REPEATED frame numbers (tennis: several crops of one frame), gaps, leading empty
    frames and a single-frame video, written with the reference's store_pickle and densified by the reference's own
    load_embs / group_by_frame (action_dataset/load.py:16-64), raw and row-normalised."""
    _, _, _, _, ref_io, ref_load = ref
    rdir = os.path.join(OUT, "reader")
    os.makedirs(rdir, exist_ok=True)
    rng = np.random.default_rng(77)
    vids = {"rally_a": ([2, 2, 3, 7, 7, 7, 9], (16,)),            # 1-D embeddings, a frame seen 2x and 3x, gaps of 4 and 2
            "rally_b": ([4, 6, 6, 11, 12, 12], (2, 16)),           # [K, D] embeddings, four leading empty frames
            "one": ([5], (16,)),                                    # a single embedded frame
            "zero_row": ([0, 3], (16,))}                            # an all-zero embedding: normalize_rows leaves it alone
    for name, (frames, shape) in vids.items():
        embs = [(int(f), rng.standard_normal(shape).astype(np.float32), {}) for f in frames]
        if name == "zero_row":
            embs[0] = (0, np.zeros(shape, np.float32), {})
        ref_io.store_pickle(os.path.join(rdir, name + ".emb.pkl"), embs)
    with open(os.path.join(rdir, "not_an_embedding.txt"), "w") as fp:      # load_embs must skip other files
        fp.write("x\n")
    res = {}
    for norm in (False, True):
        d = ref_load.load_embs(rdir, norm)
        assert sorted(d) == sorted(vids)
        for name, (dense, mask) in d.items():
            res["dense/%d/%s" % (norm, name)] = dense
            res["mask/%d/%s" % (norm, name)] = mask
    return res


WELLCOND_ARCHS = ["resnet18", "resnet34", "resnet50"]
WC_SAMPLES = 512


def wellcond_case(ref, arch):
    """Gradients of the REFERENCE student (models/rgb.py:46-70 + models/module.py ResNet, train mode, sum-MSE:
    train_vpd_model.py:79-91) in the well-conditioned regime of tests/test_model_gpu.py: reference initialisation
    (oracle.reference_init_state_dict, seed 3) with the last BatchNorm gamma of every residual branch scaled to 0.1, summed over
    three 8-crop batches, each from fresh running statistics.  Stored: WC_SAMPLES evenly spaced elements of every parameter's
    summed gradient (sample_idx), its norm, and the three losses -- the HIP gradients are compared with the reference's
    OWN numbers here, not with the oracle's restatement of them."""
    _, ref_rgb, _, ref_train, _, _ = ref
    sd = O.reference_init_state_dict(arch, 5, 32, 3)
    last = ".bn3.weight" if O.arch_expansion(arch) == 4 else ".bn2.weight"
    for k in sd:
        if k.endswith(last):
            sd[k] = sd[k] * 0.1
    out = {"meta": json.dumps(dict(arch=arch, c_in=5, emb_dim=32, n=8, hw=128, batches=3, init_seed=3, last_bn_scale=0.1,
                                   crop_seeds=[5, 15, 25], target_seeds=[6, 16, 26], samples=WC_SAMPLES))}
    acc, losses = {}, []
    for b in range(3):
        enc = ref_rgb.RGBF_EmbeddingModel(arch, 32, True, "cpu")
        enc.load_state_dict(sd)
        enc.train()
        img, tgt = O.synthetic_crops(8, 5, 128, 5 + 10 * b), O.synthetic_targets(8, 32, False, 6 + 10 * b)
        loss = torch.nn.functional.mse_loss(enc(img), tgt, reduction="sum")
        loss.backward()
        losses.append(float(loss.item()))
        for k, q in enc.named_parameters():
            acc[k] = acc.get(k, 0.0) + q.grad.detach().double()
    out["losses"] = np.asarray(losses, np.float64)
    for k, g in acc.items():
        flat = g.reshape(-1)
        out["gnorm/" + k] = np.float64(flat.norm().item())
        out["gsamp/" + k] = flat[torch.from_numpy(sample_idx(flat.numel(), WC_SAMPLES))].numpy().astype(np.float32)
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = _import_reference()
    only = os.environ.get("GOLDEN_ONLY")        # comma-separated case names: regenerate just those
    for i, case in enumerate(CASES):
        if only and case[0] not in only.split(","):
            continue
        out = run_case(ref, *case, seed=100 + 10 * i)
        np.savez_compressed(os.path.join(OUT, case[0] + ".npz"), **out)
        print("wrote", case[0], "loss", float(out["loss_train"]), "traj", out["epoch_traj"])
    for i, case in enumerate(FULLSIZE_CASES):      # only on request: GOLDEN_ONLY=<names> (minutes of CPU time, GBs of memory)
        if not only or case[0] not in only.split(","):
            continue
        out = run_case(ref, *case[:8], seed=500 + 10 * i, level=case[8], norm=case[9])
        np.savez_compressed(os.path.join(OUT, case[0] + ".npz"), **out)
        print("wrote", case[0], "loss", float(out.get("loss_train", float("nan"))))
    if not only or "init" in only.split(","):
        with open(os.path.join(OUT, "init_stats.json"), "w") as fp:
            json.dump(init_case(ref), fp, indent=0, sort_keys=True)
        print("wrote init_stats.json")
    for arch in WELLCOND_ARCHS:
        if not only or "wellcond" in only.split(","):
            np.savez_compressed(os.path.join(OUT, "wc_grads_%s.npz" % arch), **wellcond_case(ref, arch))
            print("wrote wc_grads_" + arch)
    if not only or "reader" in only.split(","):
        np.savez_compressed(os.path.join(OUT, "reader_case.npz"), **reader_case(ref))
        print("wrote reader_case")
    if not only:
        np.savez_compressed(os.path.join(OUT, "adamw_injected.npz"), **adamw_case(ref))
        np.savez_compressed(os.path.join(OUT, "format_case.npz"), **format_case(ref))
    print("done ->", OUT)


if __name__ == "__main__":
    main()
