"""CPU oracle for the VPD student train/apply path.  TEST INFRASTRUCTURE ONLY.

This file is a fp32 CPU restatement of the arithmetic of the reference's
student path.  It is the *checker* for the HIP product in ``vpd_amd/``: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``vpd_amd/`` imports it, and the product never
falls back to it.

Pinning: the reference repository holds no tests, golden vectors or fixtures
for this path, and the BasicBlock arithmetic lives in
un-vendored, un-pinned torchvision (reference ``requirements.txt:2``).  The
oracle is therefore pinned against outputs of the reference *itself* run in the
authoring container: ``oracle/gen_golden.py`` imports the reference's
``RGBF_EmbeddingModel`` / ``ModelTrainer`` / ``FCNet`` / ``step`` (with stubs
for the three missing third-party imports), drives them on seeded inputs and
commits the results as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this file against those vectors.

The network is stated functionally over a flat ``state_dict`` (the reference's
key names) rather than as an ``nn.Module`` tree.  Reference anchors:

* topology / forward order ......... models/module.py:56-69, :88-110, :112-127
* BasicBlock ....................... torchvision (conv3x3-BN-ReLU-conv3x3-BN-add-ReLU),
                                     used via models/module.py:6, :19-21
* 5-channel stem + fc swap ......... models/rgb.py:8-43
* embed() contract ................. models/rgb.py:72-86
* motion head (FCNet) .............. models/module.py:133-156, train_vpd_model.py:61-65
* loss ............................. train_vpd_model.py:87  (sum-MSE, no 1/B)
* step order ....................... models/util.py:50-58
* optimizer ........................ train_vpd_model.py:100-105 (AdamW, torch defaults)
* epoch() return value ............. train_vpd_model.py:67-98
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# models/module.py:17-32 -- ENCODER_ARCH: BasicBlock archs and the Bottleneck archs
ARCH_LAYERS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3),
               "resnet101": (3, 4, 23, 3), "wide_resnet50_2": (3, 4, 6, 3), "wide_resnet101_2": (3, 4, 23, 3)}
# Bottleneck (torchvision, used via models/module.py:6, :22-31): expansion 4; the 3x3 conv carries the stride
# ("ResNet v1.5"); width = planes * (width_per_group / 64), width_per_group 64 or 128 (models/module.py:26-31)
ARCH_BOTTLENECK_BASE_WIDTH = {"resnet50": 64, "resnet101": 64, "wide_resnet50_2": 128, "wide_resnet101_2": 128}
STAGE_WIDTH = (64, 128, 256, 512)


def arch_expansion(arch: str) -> int:
    return 4 if arch in ARCH_BOTTLENECK_BASE_WIDTH else 1
BN_EPS = 1e-5          # torch BatchNorm2d default, models/module.py:41-43
BN_MOMENTUM = 0.1
ADAMW_DEFAULTS = dict(beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01)


# ---------------------------------------------------------------------------
# state_dict schema: key order is the reference's module order
# ---------------------------------------------------------------------------
def encoder_schema(arch: str, c_in: int, emb_dim: int) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """key -> (shape, kind); kind in {conv, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, fc_w, fc_b}."""
    sch: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    def bn(prefix: str, c: int):
        sch[prefix + ".weight"] = ((c,), "bn_w")
        sch[prefix + ".bias"] = ((c,), "bn_b")
        sch[prefix + ".running_mean"] = ((c,), "bn_rm")
        sch[prefix + ".running_var"] = ((c,), "bn_rv")
        sch[prefix + ".num_batches_tracked"] = ((), "bn_nbt")

    sch["resnet.conv1.weight"] = ((64, c_in, 7, 7), "conv")
    bn("resnet.bn1", 64)
    inplanes = 64
    exp = arch_expansion(arch)
    for li, (nblk, planes) in enumerate(zip(ARCH_LAYERS[arch], STAGE_WIDTH), start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            p = "resnet.layer%d.%d" % (li, bi)
            if exp == 1:
                sch[p + ".conv1.weight"] = ((planes, inplanes, 3, 3), "conv")
                bn(p + ".bn1", planes)
                sch[p + ".conv2.weight"] = ((planes, planes, 3, 3), "conv")
                bn(p + ".bn2", planes)
            else:
                width = planes * ARCH_BOTTLENECK_BASE_WIDTH[arch] // 64
                sch[p + ".conv1.weight"] = ((width, inplanes, 1, 1), "conv")
                bn(p + ".bn1", width)
                sch[p + ".conv2.weight"] = ((width, width, 3, 3), "conv")
                bn(p + ".bn2", width)
                sch[p + ".conv3.weight"] = ((planes * exp, width, 1, 1), "conv")
                bn(p + ".bn3", planes * exp)
            if stride != 1 or inplanes != planes * exp:
                sch[p + ".downsample.0.weight"] = ((planes * exp, inplanes, 1, 1), "conv")
                bn(p + ".downsample.1", planes * exp)
            inplanes = planes * exp
    sch["resnet.fc.weight"] = ((emb_dim, 512 * exp), "fc_w")
    sch["resnet.fc.bias"] = ((emb_dim,), "fc_b")
    return sch


def decoder_schema(emb_dim: int, hidden=(128, 128)) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """FCNet(emb_dim, [128,128], 2*emb_dim, dropout=0): Linear at Sequential idx 0,2,5."""
    sch: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    dims = [emb_dim, *hidden, 2 * emb_dim]
    for idx, (i, o) in zip((0, 2, 5), zip(dims[:-1], dims[1:])):
        sch["layers.%d.weight" % idx] = ((o, i), "fc_w")
        sch["layers.%d.bias" % idx] = ((o,), "fc_b")
    return sch


def procedural_state_dict(schema, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Seeded non-trivial weights (too big to commit): every tensor, in key
    order, from one RandomState -- recipe fixed by SURVEY 8c."""
    rs = np.random.RandomState(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, (shape, kind) in schema.items():
        if kind == "conv":
            co, _, kh, kw = shape
            a = rs.standard_normal(shape) * math.sqrt(2.0 / (co * kh * kw))
        elif kind == "bn_w":
            a = rs.uniform(0.5, 1.5, shape)
        elif kind in ("bn_b", "bn_rm"):
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            a = rs.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        elif kind == "fc_w":
            bound = 1.0 / math.sqrt(shape[1])
            a = rs.uniform(-bound, bound, shape)
        elif kind == "fc_b":
            a = rs.uniform(-0.04, 0.04, shape)
        else:  # pragma: no cover
            raise KeyError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def reference_init_state_dict(arch: str, c_in: int, emb_dim: int, seed: int):
    """Reference initialisation semantics (Appendix A): kaiming fan_out convs
    (models/module.py:71-73), BN gamma=1 beta=0 (:74-76), 5-ch stem = channel
    mean of the 3-ch kernel (models/rgb.py:19-23), nn.Linear default for fc."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, (shape, kind) in encoder_schema(arch, c_in, emb_dim).items():
        if kind == "conv":
            co, ci, kh, kw = shape
            std = math.sqrt(2.0 / (co * kh * kw))
            if key == "resnet.conv1.weight" and c_in != 3:      # add_flow_to_model: channel mean expanded (5; 6 = configs[2])
                w3 = torch.randn((co, 3, kh, kw), generator=g) * std
                sd[key] = w3.mean(dim=1, keepdim=True).expand(shape).contiguous()
            else:
                sd[key] = torch.randn(shape, generator=g) * std
        elif kind in ("bn_w", "bn_rv"):
            sd[key] = torch.ones(shape)
        elif kind in ("bn_b", "bn_rm"):
            sd[key] = torch.zeros(shape)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
        else:
            fan_in = 512 * arch_expansion(arch)      # fc.in_features (models/module.py:69): 2048 for Bottleneck students
            bound = 1.0 / math.sqrt(fan_in)
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return sd


def trainable_keys(schema) -> List[str]:
    return [k for k, (_, kind) in schema.items() if kind in ("conv", "bn_w", "bn_b", "fc_w", "fc_b")]


# ---------------------------------------------------------------------------
# forward
# ---------------------------------------------------------------------------
def _bn(sd, prefix, x, train, taps=None):
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train:
        if taps is not None:
            with torch.no_grad():
                taps[prefix] = (x.mean(dim=(0, 2, 3)).clone(),
                                x.var(dim=(0, 2, 3), unbiased=False).clone())
        y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                         True, BN_MOMENTUM, BN_EPS)
        nbt = sd.get(prefix + ".num_batches_tracked")
        if nbt is not None:
            nbt += 1
        return y
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                        False, BN_MOMENTUM, BN_EPS)


class _RoundBf16(torch.autograd.Function):
    """Value AND gradient rounded to bf16 (round-to-nearest-even): models a tensor that the
    HIP path stores in bf16 in both directions (activation forward, its gradient backward)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _ScaleGrad(torch.autograd.Function):
    """Identity forward, gradient scaled by `s`: a deliberately WRONG backward, used only by the tests that prove the
    gradient gates would notice an orchestration error of that size (GRAD_FAULT below)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


# test hook: {"resnet.layerL.B": scale} multiplies the gradient flowing through that block's identity path
GRAD_FAULT: Dict[str, float] = {}


def _rb(x, on):
    return _RoundBf16.apply(x) if on else x


def _wq(w, on):
    """bf16 weight shadow: rounded value, straight-through gradient (fp32 master gets the fp32 wgrad)."""
    return w + (w.detach().bfloat16().float() - w.detach()) if on else w


def encoder_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, arch: str,
                    train: bool, taps: Optional[dict] = None, emulate_bf16: bool = False) -> torch.Tensor:
    """f32[N,C,H,W] -> f32[N,D].  models/module.py:112-127 order; train-mode BN
    mutates running stats in ``sd`` exactly like nn.BatchNorm2d.

    emulate_bf16=True restates the PRECISION of the HIP path on top of the same
    algorithm (bf16 conv operands and stored activations / activation gradients,
    fp32 accumulation, statistics, BN arithmetic, head and loss), so that tests
    can separate rounding noise from kernel errors."""
    q = emulate_bf16
    x = _rb(x, q)
    h = _rb(F.conv2d(x, _wq(sd["resnet.conv1.weight"], q), None, stride=2, padding=3), q)
    h = _rb(F.relu(_bn(sd, "resnet.bn1", h, train, taps)), q)
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for li, nblk in enumerate(ARCH_LAYERS[arch], start=1):
        for bi in range(nblk):
            p = "resnet.layer%d.%d" % (li, bi)
            stride = 2 if (bi == 0 and li > 1) else 1
            if (p + ".conv3.weight") in sd:      # Bottleneck: 1x1 -> 3x3 (stride) -> 1x1 (x4)
                o = _rb(F.conv2d(h, _wq(sd[p + ".conv1.weight"], q), None, stride=1), q)
                o = _rb(F.relu(_bn(sd, p + ".bn1", o, train, taps)), q)
                o = _rb(F.conv2d(o, _wq(sd[p + ".conv2.weight"], q), None, stride=stride, padding=1), q)
                o = _rb(F.relu(_bn(sd, p + ".bn2", o, train, taps)), q)
                o = _rb(F.conv2d(o, _wq(sd[p + ".conv3.weight"], q), None, stride=1), q)
                o = _bn(sd, p + ".bn3", o, train, taps)
            else:                                # BasicBlock
                o = _rb(F.conv2d(h, _wq(sd[p + ".conv1.weight"], q), None, stride=stride, padding=1), q)
                o = _rb(F.relu(_bn(sd, p + ".bn1", o, train, taps)), q)
                o = _rb(F.conv2d(o, _wq(sd[p + ".conv2.weight"], q), None, stride=1, padding=1), q)
                o = _bn(sd, p + ".bn2", o, train, taps)
            if (p + ".downsample.0.weight") in sd:
                idn = _rb(F.conv2d(h, _wq(sd[p + ".downsample.0.weight"], q), None, stride=stride), q)
                idn = _bn(sd, p + ".downsample.1", idn, train, taps)
            else:
                idn = h
            if p in GRAD_FAULT:
                idn = _ScaleGrad.apply(idn, GRAD_FAULT[p])
            h = _rb(F.relu(o + idn), q)
    h = h.mean(dim=(2, 3))                     # AdaptiveAvgPool2d((1,1)) + flatten
    return F.linear(h, sd["resnet.fc.weight"], sd["resnet.fc.bias"])


def decoder_forward(dsd: Dict[str, torch.Tensor], e: torch.Tensor) -> torch.Tensor:
    """FCNet D->128->128->2D, ReLU, Dropout(p=0) (identity).  models/module.py:139-156."""
    h = F.relu(F.linear(e, dsd["layers.0.weight"], dsd["layers.0.bias"]))
    h = F.relu(F.linear(h, dsd["layers.2.weight"], dsd["layers.2.bias"]))
    return F.linear(h, dsd["layers.5.weight"], dsd["layers.5.bias"])


def sum_mse(e: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """train_vpd_model.py:87 -- F.mse_loss(reduction='sum'): no 1/B, no 1/D."""
    d = e - t
    return (d * d).sum()


def embed(sd, x, arch: str, use_flow: bool) -> np.ndarray:
    """models/rgb.py:72-86: ndarray/tensor in, 3-D promoted, channel assert,
    eval-mode, numpy f32 out.  use_flow may be an int: the explicit input-channel count of the 6-channel variant
    (BASELINE configs[2]; the reference hard-codes 5)."""
    if not isinstance(use_flow, bool) and isinstance(use_flow, int):
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(np.asarray(x), dtype=torch.float32)
        if x.dim() == 3:
            x = x.unsqueeze(0)
        assert x.shape[1] == use_flow, "Wrong number of channels"
        with torch.no_grad():
            return encoder_forward(sd, x.float(), arch, train=False).numpy()
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(np.asarray(x), dtype=torch.float32)
    if x.dim() == 3:
        x = x.unsqueeze(0)
    if use_flow:
        assert x.shape[1] == 5, "Wrong number of channels for RGB + flow"
    else:
        assert x.shape[1] == 3, "Wrong number of channels for RGB"
    with torch.no_grad():
        return encoder_forward(sd, x.float(), arch, train=False).numpy()


# ---------------------------------------------------------------------------
# optimizer: AdamW, torch defaults (train_vpd_model.py:104; Appendix A)
# ---------------------------------------------------------------------------
class AdamWState:
    def __init__(self, params: "OrderedDict[str, torch.Tensor]"):
        self.t = 0
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())


def adamw_update(params, grads, st: AdamWState, lr: float, beta1=0.9, beta2=0.999,
                 eps=1e-8, weight_decay=0.01) -> None:
    """In place.  p*=1-lr*wd; m,v EMA; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    st.t += 1
    bc1 = 1.0 - beta1 ** st.t
    bc2 = 1.0 - beta2 ** st.t
    with torch.no_grad():
        for k, p in params.items():
            g = grads[k]
            p.mul_(1.0 - lr * weight_decay)
            st.m[k].lerp_(g, 1.0 - beta1)      # torch.optim's form of b1*m + (1-b1)*g
            st.v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(st.m[k], denom, value=-(lr / bc1))


# ---------------------------------------------------------------------------
# the train step and the epoch loop
# ---------------------------------------------------------------------------
class StudentOracle:
    """Holds an encoder (+ optional motion decoder) state_dict and restates
    ModelTrainer.epoch / get_optimizer / step on CPU fp32."""

    def __init__(self, arch: str, c_in: int, emb_dim: int, motion: bool,
                 enc_sd: Dict[str, torch.Tensor], dec_sd: Optional[Dict[str, torch.Tensor]] = None):
        self.arch, self.c_in, self.emb_dim, self.motion = arch, c_in, emb_dim, motion
        self.enc = OrderedDict((k, v.clone()) for k, v in enc_sd.items())
        self.dec = OrderedDict((k, v.clone()) for k, v in dec_sd.items()) if motion else None
        self.enc_keys = trainable_keys(encoder_schema(arch, c_in, emb_dim))
        self.dec_keys = list(self.dec.keys()) if motion else []
        self.opt: Optional[AdamWState] = None
        self.lr = None

    # -- parameter views ---------------------------------------------------
    def params(self) -> "OrderedDict[str, torch.Tensor]":
        p = OrderedDict(("enc." + k, self.enc[k]) for k in self.enc_keys)
        for k in self.dec_keys:
            p["dec." + k] = self.dec[k]
        return p

    def get_optimizer(self, lr: float):
        self.opt = AdamWState(self.params())
        self.lr = lr

    # -- forward / loss / backward ------------------------------------------
    def forward_loss(self, img: torch.Tensor, target: torch.Tensor, train: bool,
                     need_grad: bool, taps: Optional[dict] = None, emulate_bf16: bool = False):
        ps = self.params()
        if need_grad:
            for p in ps.values():
                p.requires_grad_(True)
                p.grad = None
        ctx = torch.enable_grad() if need_grad else torch.no_grad()
        with ctx:
            emb = encoder_forward(self.enc, img, self.arch, train, taps, emulate_bf16)
            out = decoder_forward(self.dec, emb) if self.motion else emb
            loss = sum_mse(out, target)
        grads = None
        if need_grad:
            loss.backward()
            grads = OrderedDict((k, p.grad.detach().clone()) for k, p in ps.items())
            for p in ps.values():
                p.requires_grad_(False)
                p.grad = None
        return float(loss.detach()), emb.detach(), out.detach(), grads

    def train_step(self, img, target, taps=None):
        """loss.backward(); optimizer.step(); optimizer.zero_grad()  (models/util.py:50-58)."""
        loss, emb, out, grads = self.forward_loss(img, target, train=True, need_grad=True, taps=taps)
        adamw_update(self.params(), grads, self.opt, self.lr, **{
            "beta1": 0.9, "beta2": 0.999, "eps": 1e-8, "weight_decay": 0.01})
        return loss, emb, out, grads

    def epoch(self, batches: Iterable[dict], train: bool) -> float:
        """train_vpd_model.py:67-98 -> sum of per-batch sum-MSE / number of crops."""
        tot, n = 0.0, 0
        for b in batches:
            img, tgt = b["img"], b["emb"]
            if train:
                loss, *_ = self.train_step(img, tgt)
            else:
                loss, *_ = self.forward_loss(img, tgt, train=False, need_grad=False)
            tot += loss
            n += img.shape[0]
        return tot / n


# ---------------------------------------------------------------------------
# seeded synthetic inputs in the reference's value ranges (SURVEY 8d;
# vpd_dataset/common.py:52-69, raft/flow.py:80-84)
# ---------------------------------------------------------------------------
DIVING48_MEAN_STD = ((0.3411329922282787, 0.46349889258964044, 0.5162481674015696),
                     (0.16302619019820488, 0.17092395707914718, 0.19266662199338647))


# figure-skating crops (BASELINE configs[3]): vpd_dataset/common.py:19-22
FS_MEAN_STD = ((0.5747710337842444, 0.5644043210903272, 0.6334494151377134),
               (0.21349823115367886, 0.21827191146692457, 0.20393919008463163))


def synthetic_crops(n: int, c_in: int, hw: int, seed: int, mean_std=None) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    rgb = rs.randint(0, 256, size=(n, 3, hw, hw)).astype(np.float32) / 255.0
    mean_std = mean_std or DIVING48_MEAN_STD
    mean = np.asarray(mean_std[0], np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(mean_std[1], np.float32).reshape(1, 3, 1, 1)
    x = (rgb - mean) / std
    if c_in > 3:
        fl = np.clip(np.round(124 + 12 * rs.standard_normal((n, c_in - 3, hw, hw))), 0, 255)
        x = np.concatenate([x, fl.astype(np.float32) / 255.0 - 0.5], axis=1)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


def synthetic_crops_u8(n: int, c_in: int, hw: int, seed: int):
    """The decoded u8 frames synthetic_crops(n, c_in, hw, seed) is made of (same draws): rgb u8 [n, hw, hw, 3] and, for
    c_in = 5, flow u8 [n, hw, hw, 2] -- what the crop PNGs hold before vpd_dataset/common.py:52-69 turns them into floats."""
    rs = np.random.RandomState(seed)
    rgb = rs.randint(0, 256, size=(n, 3, hw, hw)).astype(np.uint8)
    flow = None
    if c_in > 3:
        flow = np.clip(np.round(124 + 12 * rs.standard_normal((n, c_in - 3, hw, hw))), 0, 255).astype(np.uint8)
        flow = torch.from_numpy(np.ascontiguousarray(flow.transpose(0, 2, 3, 1)))
    return torch.from_numpy(np.ascontiguousarray(rgb.transpose(0, 2, 3, 1))), flow


def synthetic_targets(n: int, emb_dim: int, motion: bool, seed: int) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    t = rs.standard_normal((n, emb_dim)).astype(np.float32)
    if motion:   # vpd_dataset/single_frame.py:256-258: [t, t - t_prev]
        tp = rs.standard_normal((n, emb_dim)).astype(np.float32)
        t = np.concatenate([t, t - tp], axis=1)
    return torch.from_numpy(t)
