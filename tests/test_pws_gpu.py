"""conv3x3_pws_kernel (persistent blocks, LDS flag hand-off; vpd_amd/csrc/conv_pws.h) against plain PyTorch fp32 on the
same bf16-rounded operands, through the C ABI (vpd_op_conv2d).  Each case runs in a child process with VPD_PWS_BLOCKS set,
so that a block walks MANY tiles (ring wrap-around, halo hand-over between tiles, ragged last tile, statistics kept in
registers across tiles) on problems small enough for a CPU reference, and once more with VPD_PWS=0 (conv3x3_ws_kernel):
the two kernels must agree bit for bit, because they add the same products in the same order.
Reference shapes: /root/reference/models/module.py:61-67 (3x3 stride-1 convs of layer2 / layer3 / layer4)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import ctypes as C, json, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, {repo!r})
from tests.test_ops_gpu import (bf16_round, from_nhwc, pack_dgrad, pack_fwd, rel_l2, run_conv, tapset, to_padded_nhwc)

n, ci, co, h, w = {case!r}
g = torch.Generator().manual_seed(n + ci + h)
x = bf16_round(torch.randn(n, ci, h, w, generator=g))
wt = bf16_round(torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5)
torch.set_num_threads(8)
ref = F.conv2d(x, wt, None, stride=1, padding=1)
dz = bf16_round(torch.randn(ref.shape, generator=g))
dxr = F.conv_transpose2d(dz, wt, None, stride=1, padding=1)
out = {{}}
xp = to_padded_nhwc(x, 1, 1, 1, 1)
taps = tapset(3, 3, 0, 1, 0, 1, 0, 3, 1)
y, stats = run_conv(xp, pack_fwd(wt), n, h + 2, w + 2, ci, h, w, 0, h, w, 1, 0, 0, 1, ci, co, taps, want_stats=True)
got = from_nhwc(y, n, h, w, co, 0)
out["fwd"] = rel_l2(got, ref)
s = stats.sum(dim=0).cpu()
out["sum"] = rel_l2(s[0], got.sum(dim=(0, 2, 3)))
out["sumsq"] = rel_l2(s[1], (got * got).sum(dim=(0, 2, 3)))
out["y_crc"] = int(y.view(torch.int16).to(torch.int64).sum().item())
# data gradient: plain store, then accumulate on top
dzp = to_padded_nhwc(dz, 1, 1, 1, 1)
wd = pack_dgrad(wt)
dx = torch.zeros(n * h * w * ci, dtype=torch.bfloat16, device="cuda")
tapsd = tapset(3, 3, 2, -1, 2, -1, 0, 3, 1)
run_conv(dzp, wd, n, h + 2, w + 2, co, h, w, 0, h, w, 1, 0, 0, 1, co, ci, tapsd, y=dx)
out["dgrad"] = rel_l2(from_nhwc(dx, n, h, w, ci, 0), dxr)
out["dx_crc"] = int(dx.view(torch.int16).to(torch.int64).sum().item())
run_conv(dzp, wd, n, h + 2, w + 2, co, h, w, 0, h, w, 1, 0, 0, 1, co, ci, tapsd, y=dx, accumulate=1)
out["acc"] = rel_l2(from_nhwc(dx, n, h, w, ci, 0), 2 * dxr)
# a second launch on the same buffers must reproduce the first bit for bit (flags start from zero every launch)
y2, _ = run_conv(xp, pack_fwd(wt), n, h + 2, w + 2, ci, h, w, 0, h, w, 1, 0, 0, 1, ci, co, taps, want_stats=True)
out["repeat"] = bool(torch.equal(y, y2))
print("RESULT " + json.dumps(out))
"""

# (crops, Ci, Co, H, W): the tile class each shape selects is named in the id (vpd_conv_kernel_class)
CASES = {
    "c1_256x128_layer2": (203, 128, 128, 16, 16),          # 203 tiles of one 16 x 16 image each
    "c6_256x64_layer3_ragged": (203, 256, 256, 8, 8),      # 256-pixel tiles = 4 images: 50.75 tiles x 4 channel tiles
    "c2_128x128_layer4_ragged": (403, 512, 512, 4, 4),     # 128-pixel tiles = 8 images: 50.4 tiles x 4 channel tiles
    "c3_128x64_layer4_small": (37, 512, 512, 4, 4),        # < 200 tiles of 128 x 128: 128 x 64 tiles, 4.6 x 8
    "c3_128x64_one_chunk_pair": (21, 128, 128, 4, 4),      # two 64-channel chunks only: 18 K-steps per tile
}


def _run(case, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", _CHILD.format(repo=REPO, case=CASES[case])], env=env, capture_output=True,
                       text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


# every tile class with few blocks (many tiles per block: ring and halo wrap-around); one block per CU for the two classes of
# the BASELINE step's layer2 / layer3 launches (the full-size tests run the rest that way)
_PWS_RUNS = [(c, "24") for c in CASES] + [("c1_256x128_layer2", "0"), ("c6_256x64_layer3_ragged", "0")]


@pytest.mark.parametrize("case,blocks", _PWS_RUNS, ids=["%s-%s" % (c, "few_blocks" if b == "24" else "device_blocks") for c, b in _PWS_RUNS])
def test_pws_conv_matches_reference_and_old_kernel(case, blocks):
    new = _run(case, {"VPD_PWS": "1", "VPD_PWS_BLOCKS": blocks})
    assert new["fwd"] < 4e-3 and new["dgrad"] < 4e-3 and new["acc"] < 8e-3, new
    assert new["sum"] < 1e-4 and new["sumsq"] < 1e-4, new
    assert new["repeat"], new
    old = _run(case, {"VPD_PWS": "0"})
    assert old["fwd"] < 4e-3 and old["dgrad"] < 4e-3
    # same products, same order of additions: identical bf16 outputs
    assert new["y_crc"] == old["y_crc"] and new["dx_crc"] == old["dx_crc"], (new, old)


def test_train_steps_are_bit_identical_with_the_pool_gradient_folded():
    """VPD_POOLBWD_FOLD=0 (the avgpool_bwd launch instead of the last BatchNorm backward producing d(out) itself) must not change a
    bit of the train step: losses of three steps at 256 crops and a SHA-256 over all parameters and BatchNorm buffers, one child
    process per setting (the switch is read once)."""
    outs = {}
    for name, extra in (("default", {}), ("pool_launch", {"VPD_POOLBWD_FOLD": "0"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "step_digest.py"), "--steps", "3"], env=env,
                           capture_output=True, text=True, timeout=900, cwd=REPO)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("losses ")][-1]
    assert outs["pool_launch"] == outs["default"], outs


def test_bottleneck_student_on_the_streaming_1x1_kernel_matches_the_gather_kernel(tmp_path):
    """conv1x1_stream_kernel takes layer1's 1x1 convolutions of a ResNet-50 only from 64 crops per step up -- more than the golden
    cases hold.  The train-mode loss and the eval-mode embedding of a 64-crop batch on freshly initialised weights, with it
    (default) and without (VPD_CONV1X1_STREAM=0: conv_igemm_kernel, which the goldens pin).  Its outputs are bit-identical to the
    gather kernel's and its statistics agree to 1e-8 (tests/test_ops_gpu.py covers both against torch): eval embeddings must be
    EQUAL, the train loss (batch statistics at 1e-8 flip a few bf16 roundings downstream) within 2e-3."""
    import numpy as np
    outs = {}
    for name, extra in (("stream", {}), ("gather", {"VPD_CONV1X1_STREAM": "0"})):
        env = dict(os.environ, **extra)
        npy = str(tmp_path / (name + ".npy"))
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "step_digest.py"), "--arch", "resnet50", "--batch", "64",
                            "--steps", "1", "--eval-first", "--eval-out", npy], env=env, capture_output=True, text=True, timeout=900,
                           cwd=REPO)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("losses ")][-1]
        outs[name] = (float(line.split()[1]), np.load(npy))
    (ls, es), (lg, eg) = outs["stream"], outs["gather"]
    assert abs(ls - lg) <= 2e-3 * abs(lg), (ls, lg)
    assert np.array_equal(es, eg), float(np.abs(es - eg).max())
