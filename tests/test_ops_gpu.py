"""Operator-level parity of the HIP kernels (through the C ABI) against plain
PyTorch fp32 on the CPU, on seeded inputs.  Operands are bf16 (inputs are
rounded to bf16 first, so the comparison isolates accumulation order and the
bf16 rounding of the output): tolerance rel-L2 <= 4e-3, stated per test."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REL_TOL = 4e-3


def _lib():
    from vpd_amd._lib import lib
    return lib()


def _check(rc):
    from vpd_amd._lib import check
    check(rc, "op")


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def ptr(t):
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def to_padded_nhwc(x, pad_t, pad_l, pad_b, pad_r, slack=0):
    """f32 NCHW (cpu) -> bf16 padded NHWC on the GPU (flat, with `slack` extra zero elements)."""
    n, c, h, w = x.shape
    buf = torch.zeros(n, h + pad_t + pad_b, w + pad_l + pad_r, c, dtype=torch.bfloat16)
    buf[:, pad_t:pad_t + h, pad_l:pad_l + w, :] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    flat = torch.zeros(buf.numel() + slack, dtype=torch.bfloat16)
    flat[:buf.numel()] = buf.flatten()
    return flat.cuda()


def from_nhwc(flat, n, hp, wp, c, pad):
    t = flat[: n * hp * wp * c].view(n, hp, wp, c).float().cpu()
    if pad:
        t = t[:, pad:-pad, pad:-pad, :]
    return t.permute(0, 3, 1, 2).contiguous()


def tapset(*v):
    return (C.c_int * 9)(*v)


def run_conv(xp, wp, n, xHp, xWp, xC, yH, yW, ypad, Hs, Ws, osub, oph, opw, istr, Kc, Co, taps, y=None,
             accumulate=0, want_stats=False, yC=None):
    L = _lib()
    yC = yC or Co
    yHp, yWp = yH + 2 * ypad, yW + 2 * ypad
    if y is None:
        y = torch.zeros(n * yHp * yWp * yC, dtype=torch.bfloat16, device="cuda")
    stats = None
    if want_stats:
        stats = torch.zeros(16, 2, Co, dtype=torch.float64, device="cuda")     # VPD_STAT_ROWS accumulator rows
    _check(L.vpd_op_conv2d(ptr(xp), ptr(wp), ptr(y), ptr(stats) if stats is not None else None, n, xHp, xWp, xC,
                           yHp, yWp, yC, ypad, Hs, Ws, osub, oph, opw, istr, Kc, Co, taps, accumulate, stream()))
    torch.cuda.synchronize()
    return y, stats


def pack_fwd(w):
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci).contiguous().to(torch.bfloat16).cuda()


def pack_dgrad(w):
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co).contiguous().to(torch.bfloat16).cuda()


def test_tr_read_probe():
    """ds_read_b64_tr_b16 lane mapping used by the wgrad kernel."""
    L = _lib()
    tile = torch.arange(128 * 64, dtype=torch.float32).remainder(251.0).view(128, 64)
    tile = tile + torch.arange(128).view(128, 1) * 0.0   # value = (r*64+c) % 251, exactly representable
    d = tile.to(torch.bfloat16).cuda()
    out = torch.zeros(4 * 4 * 64 * 8, dtype=torch.bfloat16, device="cuda")
    _check(L.vpd_op_tr_read_probe(ptr(d), ptr(out), stream()))
    torch.cuda.synchronize()
    got = out.view(4, 4, 64, 8).float().cpu()
    exp = torch.empty(4, 4, 64, 8)
    tb = tile.to(torch.bfloat16).float()
    for wv in range(4):
        for ct in range(4):
            for l in range(64):
                for j in range(8):
                    exp[wv, ct, l, j] = tb[wv * 32 + 8 * (l >> 4) + j, ct * 16 + (l & 15)]
    assert torch.equal(got, exp)


CONV_CASES = [
    # name, N, Ci, Co, H, W, k, stride, pad
    ("l1_3x3_s1", 3, 64, 64, 16, 16, 3, 1, 1),
    ("l2_3x3_s2", 3, 64, 128, 16, 16, 3, 2, 1),
    ("l2_1x1_s2", 3, 64, 128, 16, 16, 1, 2, 0),
    ("l3_3x3_s1_ragged", 5, 256, 256, 4, 4, 3, 1, 1),
    ("l4_3x3_s1_tiny", 2, 512, 512, 2, 2, 3, 1, 1),
    ("l2_3x3_s1_big", 9, 128, 128, 16, 16, 3, 1, 1),
    # stride-2 3x3 at the three ResNet stage boundaries (halo-form weight gradient: 4 rows x 16, 8 x 8, four whole 4 x 4 images
    # per 64-pixel chunk), the last one with a ragged final chunk
    ("l2_3x3_s2_32to16", 2, 64, 128, 32, 32, 3, 2, 1),
    ("l4_3x3_s2_8to4_ragged", 5, 256, 512, 8, 8, 3, 2, 1),
    # 1x1 convolutions (Bottleneck blocks, down-sampling branches): the weight gradient runs as the centre tap of the halo kernel
    ("b1_1x1_s1_256to64", 3, 256, 64, 16, 16, 1, 1, 0),
    ("b3_1x1_s1_128to512_ragged", 5, 128, 512, 4, 4, 1, 1, 0),
    ("ds_1x1_s2_32to16", 2, 64, 128, 32, 32, 1, 2, 0),
    # the ring-GEMM kernel for deep 1x1 convs (conv1x1_ws_kernel): 256 x 128 tiles / 3 stages, 128 x 128 / 4 stages (forward),
    # 128 x 64 (the data gradient of the second case: K = 256, 1,024 output channels -> 128 x 128), ragged last tiles
    ("b_1x1_s1_512to128_t256", 50, 512, 128, 32, 32, 1, 1, 0),
    ("b_1x1_s1_1024to256_t128", 200, 1024, 256, 8, 8, 1, 1, 0),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case):
    name, n, ci, co, h, w, k, stride, pad = case
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = bf16_round(torch.randn(n, ci, h, w, generator=g))
    wt = bf16_round(torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5)
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    dz = bf16_round(torch.randn(ref.shape, generator=g))
    ref.backward(dz)

    # ---- forward (+ channel statistics) ----
    xp = to_padded_nhwc(x, 1, 1, 1, 1)
    taps = tapset(k, k, 1 - pad, 1, 1 - pad, 1, 0, k, 1)
    y, stats = run_conv(xp, pack_fwd(wt), n, h + 2, w + 2, ci, ho, wo, 0, ho, wo, 1, 0, 0, stride, ci, co, taps,
                        want_stats=True)
    got = from_nhwc(y, n, ho, wo, co, 0)
    assert rel_l2(got, ref.detach()) < REL_TOL
    s = stats.sum(dim=0).cpu()
    assert rel_l2(s[0], got.sum(dim=(0, 2, 3))) < 1e-4          # stats are over the stored bf16 values
    assert rel_l2(s[1], (got * got).sum(dim=(0, 2, 3))) < 1e-4

    # ---- data gradient ----
    dzp = to_padded_nhwc(dz, 1, 1, 1, 1)
    wd = pack_dgrad(wt)
    dx = torch.zeros(n * h * w * ci, dtype=torch.bfloat16, device="cuda")
    if stride == 1:
        taps = tapset(k, k, pad + 1, -1, pad + 1, -1, 0, k, 1)
        run_conv(dzp, wd, n, ho + 2, wo + 2, co, h, w, 0, h, w, 1, 0, 0, 1, co, ci, taps, y=dx)
    else:
        for ph in range(2):
            for pw in range(2):
                hs, ws = (h - ph + 1) // 2, (w - pw + 1) // 2
                rf, tf = (ph + pad) % 2, (pw + pad) % 2
                nr = (k - rf + 1) // 2 if rf < k else 0
                nc = (k - tf + 1) // 2 if tf < k else 0
                if nr == 0 or nc == 0:
                    continue
                taps = tapset(nr, nc, (ph + pad - rf) // 2 + 1, -1, (pw + pad - tf) // 2 + 1, -1, rf * k + tf, 2 * k, 2)
                run_conv(dzp, wd, n, ho + 2, wo + 2, co, h, w, 0, hs, ws, 2, ph, pw, 1, co, ci, taps, y=dx)
    gotdx = from_nhwc(dx, n, h, w, ci, 0)
    assert rel_l2(gotdx, xr.grad) < REL_TOL

    # accumulate flag: running the stride-1 dgrad again on top doubles the result
    if stride == 1:
        taps = tapset(k, k, pad + 1, -1, pad + 1, -1, 0, k, 1)
        run_conv(dzp, wd, n, ho + 2, wo + 2, co, h, w, 0, h, w, 1, 0, 0, 1, co, ci, taps, y=dx, accumulate=1)
        assert rel_l2(from_nhwc(dx, n, h, w, ci, 0), 2 * xr.grad) < 2 * REL_TOL

    # ---- weight gradient ----
    L = _lib()
    taps = tapset(k, k, 1 - pad, 1, 1 - pad, 1, 0, k, 1)
    slab = torch.empty(L.vpd_op_wgrad_slab_bytes() // 4, dtype=torch.float32, device="cuda")
    for use_slab in (False, True):        # generic (atomics) kernel, then the halo + slab kernel where eligible
        dw = torch.zeros(k * k, co, ci, dtype=torch.float32, device="cuda")
        _check(L.vpd_op_wgrad(ptr(dzp), ptr(xp), ptr(dw), n, ho + 2, wo + 2, co, 1, h + 2, w + 2, ci, ho, wo, stride,
                              ci, co, taps, ptr(slab) if use_slab else None, stream()))
        torch.cuda.synchronize()
        gotdw = dw.cpu().view(k, k, co, ci).permute(2, 3, 0, 1)
        assert rel_l2(gotdw, wr.grad) < REL_TOL, use_slab


def test_stem_conv_and_wgrad():
    """7x7 s2 p3 stem on a 5-channel input stored as 8-channel NHWC with a 3-pixel border:
    each kernel row is one 64-wide tap (8 column taps x 8 channels, the 8th tap / channels 5..7 zero)."""
    n, c, h, w, co = 3, 5, 32, 32, 64
    g = torch.Generator().manual_seed(7)
    x = bf16_round(torch.randn(n, c, h, w, generator=g))
    wt = bf16_round(torch.randn(co, c, 7, 7, generator=g) * (2.0 / (co * 49)) ** 0.5)
    xr, wr = x.clone(), wt.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=2, padding=3)
    ho, wo = ref.shape[2], ref.shape[3]
    dz = bf16_round(torch.randn(ref.shape, generator=g))
    ref.backward(dz)
    x8 = torch.zeros(n, 8, h, w)
    x8[:, :c] = x
    xp = to_padded_nhwc(x8, 3, 3, 3, 5, slack=256)
    wp = torch.zeros(7, co, 8, 8)
    wp[:, :, :7, :c] = wt.permute(2, 0, 3, 1)                     # [r][co][t][c]
    wp = wp.reshape(7, co, 64).to(torch.bfloat16).cuda()
    taps = tapset(7, 1, 0, 1, 0, 0, 0, 1, 0)
    y, _ = run_conv(xp, wp, n, h + 6, w + 8, 8, ho, wo, 0, ho, wo, 1, 0, 0, 2, 64, co, taps)
    assert rel_l2(from_nhwc(y, n, ho, wo, co, 0), ref.detach()) < REL_TOL

    L = _lib()
    dzd = dz.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    slab = torch.empty(L.vpd_op_wgrad_slab_bytes() // 4, dtype=torch.float32, device="cuda")
    for use_slab in (False, True):        # generic (atomics) kernel, then the raw-row stem kernel + slab
        dw = torch.zeros(7, co, 64, dtype=torch.float32, device="cuda")
        _check(L.vpd_op_wgrad(ptr(dzd), ptr(xp), ptr(dw), n, ho, wo, co, 0, h + 6, w + 8, 8, ho, wo, 2, 64, co, taps,
                              ptr(slab) if use_slab else None, stream()))
        torch.cuda.synchronize()
        got = dw.cpu().view(7, co, 8, 8)[:, :, :7, :c].permute(1, 3, 0, 2)   # -> [co][c][r][t]
        assert rel_l2(got, wr.grad) < REL_TOL, use_slab


WG128_GROUPS = {
    # one launch each: (n, H, W, Co, Ci, stride[, k]) per problem, H x W the OUTPUT size, k = 3 (default) or 1.  Different halo geometries share a launch
    # (layer3 + layer4 shapes), ragged final chunks (n * H * W not a multiple of 64), several tiles per problem, pixel splits
    # with a slab, and the stride-2 3x3 of a stage's first block (two-stage ring) beside stride-1 problems (four stages)
    "layer2_like": [(9, 16, 16, 128, 128, 1), (9, 16, 16, 128, 128, 1)],
    "layer3_and_layer4_ragged": [(5, 8, 8, 256, 256, 1), (5, 4, 4, 512, 512, 1), (7, 8, 8, 256, 128, 1)],
    "layer1_wide_rows": [(3, 32, 32, 128, 64, 1)],
    "many_chunks_split": [(40, 16, 16, 128, 64, 1), (33, 8, 8, 128, 128, 1)],
    "stride2_32to16_with_stride1": [(6, 16, 16, 128, 64, 2), (6, 16, 16, 128, 128, 1)],
    "stride2_16to8_and_8to4_ragged": [(5, 8, 8, 256, 128, 2), (5, 4, 4, 512, 256, 2), (5, 4, 4, 512, 512, 1)],
    # 1x1 convolutions as tasks of the same launch: Bottleneck shapes (128 x 128 and 128 x 64 tiles, ragged chunks, splits),
    # the stride-2 down-sampling branch, mixed with a 3x3 problem
    "one_by_one_bottleneck": [(9, 16, 16, 128, 512, 1, 1), (9, 16, 16, 512, 128, 1, 1), (5, 8, 8, 256, 64, 1, 1)],
    "one_by_one_stride2_with_3x3": [(6, 16, 16, 128, 64, 2, 1), (5, 4, 4, 512, 256, 2, 1), (6, 16, 16, 128, 128, 1, 3), (33, 8, 8, 1024, 256, 1, 1)],
}


@pytest.mark.parametrize("name", list(WG128_GROUPS), ids=list(WG128_GROUPS))
def test_wgrad128_group(name):
    """conv_wgrad128_persistent_kernel (128 x 64 tiles, persistent blocks, host-built schedule): every problem of a grouped
    launch against torch's conv2d weight gradient on the same bf16-rounded operands."""
    L = _lib()
    probs = WG128_GROUPS[name]
    g = torch.Generator().manual_seed(len(name))
    keep, refs = [], []
    dzs, xs, dws, slabs, dims = [], [], [], [], []
    probs = [pr if len(pr) == 7 else pr + (3,) for pr in probs]
    for (n, h, w, co, ci, st, ks) in probs:
        x = bf16_round(torch.randn(n, ci, st * h, st * w, generator=g))
        dz = bf16_round(torch.randn(n, co, h, w, generator=g))
        wr = torch.zeros(co, ci, ks, ks, requires_grad=True)
        F.conv2d(x, wr, None, stride=st, padding=ks // 2).backward(dz)
        refs.append(wr.grad)
        xp, dzp = to_padded_nhwc(x, 1, 1, 1, 1), to_padded_nhwc(dz, 1, 1, 1, 1)
        dw = torch.full((ks * ks, co, ci), float("nan"), dtype=torch.float32, device="cuda")      # the kernel OVERWRITES
        slab = torch.empty(max(int(L.vpd_op_wgrad128_slab_floats(co, ci)), 4), dtype=torch.float32, device="cuda")
        keep += [xp, dzp, dw, slab]
        dzs.append(dzp.data_ptr()); xs.append(xp.data_ptr()); dws.append(dw.data_ptr()); slabs.append(slab.data_ptr())
        dims += [n, h, w, co, ci, st, ks]
    k = len(probs)
    arr = lambda v: (C.c_void_p * k)(*v)
    table = torch.empty(int(L.vpd_op_wgrad128_table_bytes()), dtype=torch.uint8, device="cuda")
    _check(L.vpd_op_wgrad128_group(k, arr(dzs), arr(xs), arr(dws), arr(slabs), (C.c_int * (7 * k))(*dims), ptr(table),
                                   stream()))
    torch.cuda.synchronize()
    for i, (n, h, w, co, ci, st, ks) in enumerate(probs):
        got = keep[4 * i + 2].cpu().view(ks, ks, co, ci).permute(2, 3, 0, 1)
        assert torch.isfinite(got).all(), (name, i)
        assert rel_l2(got, refs[i]) < REL_TOL, (name, i, rel_l2(got, refs[i]))


# ---------------------------------------------------------------------------
# BatchNorm at operator level (parity soft spot 1): the whole-network gradient gates are statistical, so the
# BatchNorm passes are ALSO pinned one by one, against torch's own batch_norm in float64 on the same bf16 operands --
# the per-channel sums taken in the data gradient's epilogue (exact up to fp32 partial sums), the forward finalize + apply,
# and the backward finalize + apply (dz to the rounding of its bf16 output: a 2 % error in one coefficient is 5x the gate).
# ---------------------------------------------------------------------------
def _mask_bits(mask_bool):
    """[M][C] bool -> [M][C/8] bytes, bit j of byte (m, c8) = element (m, 8 c8 + j)."""
    m, c = mask_bool.shape
    w = (2 ** torch.arange(8, dtype=torch.int32)).view(1, 1, 8)
    return (mask_bool.view(m, c // 8, 8).to(torch.int32) * w).sum(dim=2).to(torch.uint8)


BN_SHAPES = [(4, 8, 8, 256), (2, 16, 16, 128), (3, 32, 32, 64), (5, 4, 4, 512)]


@pytest.mark.parametrize("shape", BN_SHAPES, ids=["l3", "l2", "l1", "l4"])
@pytest.mark.parametrize("residual", [False, True], ids=["plain", "residual"])
def test_batchnorm_forward_op(shape, residual):
    n, h, w, c = shape
    L = _lib()
    g = torch.Generator().manual_seed(n * 1000 + c + residual)
    z = bf16_round(torch.randn(n, c, h, w, generator=g) * 1.7 + 0.3)
    res = bf16_round(torch.randn(n, c, h, w, generator=g)) if residual else None
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    zd = z.double()
    M = n * h * w
    # the accumulator rows as a producing convolution leaves them: partial sums spread over the four rows
    parts = torch.rand(4, generator=g).double()
    parts = parts / parts.sum()
    s1, s2 = zd.sum(dim=(0, 2, 3)), (zd * zd).sum(dim=(0, 2, 3))
    rows = torch.stack([torch.stack([s1 * f, s2 * f]) for f in parts]).contiguous().cuda()
    zdev = z.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    outp = torch.zeros(n * (h + 2) * (w + 2) * c, dtype=torch.bfloat16, device="cuda")
    resp = to_padded_nhwc(res, 1, 1, 1, 1) if residual else None
    mask = torch.zeros(M * c // 8, dtype=torch.uint8, device="cuda")
    dev = lambda t: t.clone().float().cuda()
    rm, rv = dev(rm0), dev(rv0)
    mean, rstd, scale, shift = (torch.zeros(c, device="cuda") for _ in range(4))
    gd, bd = dev(gamma), dev(beta)
    _check(L.vpd_op_bn_forward(ptr(zdev), ptr(rows), ptr(gd), ptr(bd), ptr(rm), ptr(rv), ptr(mean), ptr(rstd), ptr(scale), ptr(shift),
                               ptr(resp) if residual else None, ptr(outp), ptr(mask), n, h, w, c, 1, C.c_float(0.1), C.c_float(1e-5),
                               stream()))
    torch.cuda.synchronize()
    # reference: torch's batch_norm in float64 on the same bf16 z
    rm_ref, rv_ref = rm0.double().clone(), rv0.double().clone()
    y = F.batch_norm(zd, rm_ref, rv_ref, gamma.double(), beta.double(), training=True, momentum=0.1, eps=1e-5)
    if residual:
        y = y + res.double()
    y = y.clamp_min(0)
    mu, var = zd.mean(dim=(0, 2, 3)), zd.var(dim=(0, 2, 3), unbiased=False)
    assert torch.allclose(mean.cpu().double(), mu, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rstd.cpu().double(), (var + 1e-5).rsqrt(), rtol=1e-5)
    assert torch.allclose(rm.cpu().double(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rv.cpu().double(), rv_ref, rtol=1e-5)
    assert torch.allclose(scale.cpu().double(), gamma.double() * (var + 1e-5).rsqrt(), rtol=1e-5)
    got = from_nhwc(outp, n, h + 2, w + 2, c, 1).double()
    assert rel_l2(got, y) < 3e-3                                   # bf16 output rounding (2^-9 per element)
    assert float((got - y).abs().max()) <= 2.0 ** -7 * float(y.abs().max())
    border = outp.view(n, h + 2, w + 2, c).float()
    assert float(border[:, 0].abs().max()) == 0.0 and float(border[:, :, 0].abs().max()) == 0.0      # the zero border is never written
    # the ReLU bit map is the sign of the STORED activation
    want_bits = _mask_bits((got.permute(0, 2, 3, 1).reshape(M, c) > 0))
    assert torch.equal(mask.cpu().view(M, c // 8), want_bits)


@pytest.mark.parametrize("shape", BN_SHAPES, ids=["l3", "l2", "l1", "l4"])
def test_batchnorm_backward_op_matches_autograd(shape):
    n, h, w, c = shape
    L = _lib()
    g = torch.Generator().manual_seed(7 * n + c)
    z = bf16_round(torch.randn(n, c, h, w, generator=g) * 1.3 - 0.2)
    # an incoming gradient that is CORRELATED with xhat, so that the mean(g xhat) term of the backward carries weight
    zn = (z - z.mean(dim=(0, 2, 3), keepdim=True)) / z.std(dim=(0, 2, 3), keepdim=True)
    dy = bf16_round(torch.randn(n, c, h, w, generator=g) + 0.9 * zn + 0.3)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    M = n * h * w
    zt = z.double().requires_grad_(True)
    gt = gamma.double().requires_grad_(True)
    bt = beta.double().requires_grad_(True)
    y = F.batch_norm(zt, None, None, gt, bt, training=True, eps=1e-5)
    act = y.clamp_min(0)
    (act * dy.double()).sum().backward()
    mask = (y.detach() > 0)
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(M, c)
    gm = nhwc(dy.double() * mask)
    rows = torch.zeros(4, 2, c, dtype=torch.float64)
    rows[1, 0], rows[2, 1] = gm.sum(0), (gm * nhwc(z.double())).sum(0)       # sum g, sum g * z (in different rows: they are summed)
    mu = z.double().mean(dim=(0, 2, 3))
    rstd = (z.double().var(dim=(0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    dzp = torch.zeros(n * (h + 2) * (w + 2) * c, dtype=torch.bfloat16, device="cuda")
    dgam, dbet = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dev16 = lambda t: nhwc(t).contiguous().to(torch.bfloat16).cuda()
    dyd, zd, bits = dev16(dy), dev16(z), _mask_bits(nhwc(mask)).cuda()
    rows_d, gam_d, mu_d, rs_d = rows.cuda(), gamma.float().cuda(), mu.float().cuda(), rstd.float().cuda()
    _check(L.vpd_op_bn_backward_apply(ptr(dyd), ptr(zd), ptr(bits), ptr(rows_d), ptr(gam_d), ptr(mu_d), ptr(rs_d), ptr(dzp),
                                      ptr(dgam), ptr(dbet), n, h, w, c, stream()))
    torch.cuda.synchronize()
    got = from_nhwc(dzp, n, h + 2, w + 2, c, 1).double()
    assert rel_l2(got, zt.grad) < 3e-3, rel_l2(got, zt.grad)       # bf16 output rounding; a 2 % coefficient error reads >= 1e-2
    assert rel_l2(dgam.cpu(), gt.grad) < 1e-5 and rel_l2(dbet.cpu(), bt.grad) < 1e-5
    # resolution of the gate: the same comparison against a reference whose xhat term is 2 % off fails it
    xhat = (z.double() - mu.view(1, c, 1, 1)) * rstd.view(1, c, 1, 1)
    gmm = dy.double() * mask
    off = gamma.double().view(1, c, 1, 1) * rstd.view(1, c, 1, 1) * (gmm - gmm.mean(dim=(0, 2, 3), keepdim=True)
                                                                    - 1.02 * xhat * (gmm * xhat).mean(dim=(0, 2, 3), keepdim=True))
    assert rel_l2(got, off) > 2 * 3e-3


BNSUM_CASES = [("l3_ragged", 5, 256, 256, 8, 8), ("l2", 9, 128, 128, 16, 16), ("l1", 3, 64, 64, 16, 16), ("l4", 6, 512, 512, 4, 4),
               # 1x1 (kernel size as 7th field): the Bottleneck students' conv3 / conv1 data gradients of layer3 / layer4, whose epilogue
               # (ring GEMM / gather kernel, mode 6) takes the sums of bn2 / bn1
               ("r50_l3_conv3_1x1", 32, 1024, 256, 8, 8, 1), ("r50_l4_conv3_1x1", 64, 2048, 512, 4, 4, 1),
               ("r50_l3_conv1_1x1", 8, 256, 1024, 8, 8, 1)]


@pytest.mark.parametrize("case", BNSUM_CASES, ids=[c[0] for c in BNSUM_CASES])
@pytest.mark.parametrize("accumulate", [0, 1], ids=["store", "accumulate"])
def test_conv_epilogue_batchnorm_sums(case, accumulate):
    """Epilogue modes 6 / 7: the data gradient d it stores (or adds onto the identity path's gradient) and, from the same
    registers, sum g and sum g * z of the consuming BatchNorm with g = d * mask -- checked against float64 sums over the
    kernel's OWN stored output, so only the summation is in question (fp32 partials per block, fp64 across blocks)."""
    name, n, ci, co, h, w = case[:6]
    k = case[6] if len(case) > 6 else 3
    L = _lib()
    g = torch.Generator().manual_seed(len(name) * 100 + n + accumulate)
    x = bf16_round(torch.randn(n, ci, h, w, generator=g))
    wt = bf16_round(torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5)
    M = n * h * w
    z = bf16_round(torch.randn(M, co, generator=g))
    mask = torch.rand(M, co, generator=g) > 0.45
    old = bf16_round(torch.randn(M, co, generator=g)) if accumulate else torch.zeros(M, co)
    xp = to_padded_nhwc(x, 1, 1, 1, 1)
    y = old.to(torch.bfloat16).cuda().flatten().contiguous()
    zd, bits = z.to(torch.bfloat16).cuda().contiguous(), _mask_bits(mask).cuda()
    rows = torch.zeros(4, 2, co, dtype=torch.float64, device="cuda")
    taps = tapset(3, 3, 0, 1, 0, 1, 0, 3, 1) if k == 3 else tapset(1, 1, 1, 1, 1, 1, 0, 1, 1)
    _check(L.vpd_op_conv2d_bnsums(ptr(xp), ptr(pack_fwd(wt)), ptr(y), ptr(zd), ptr(bits), ptr(rows), n, h + 2, w + 2, ci, h, w,
                                  ci, co, taps, accumulate, stream()))
    torch.cuda.synchronize()
    ref = F.conv2d(x, wt, None, padding=k // 2).permute(0, 2, 3, 1).reshape(M, co) + old
    d = y.view(M, co).float().cpu()
    assert rel_l2(d, ref) < REL_TOL
    gm = d.double() * mask
    s = rows.sum(dim=0).cpu()
    want1, want2 = gm.sum(0), (gm * z.double()).sum(0)
    assert float((s[0] - want1).abs().max()) <= 2e-5 * float(gm.abs().sum(0).max())
    assert float((s[1] - want2).abs().max()) <= 2e-5 * float((gm * z.double()).abs().sum(0).max())


# conv1x1_stream_kernel (conv_stream.hip): the persistent streaming kernel of the Bottleneck students' layer1 / layer2 1x1 convs.
# It takes a launch only with >= 2 pixel tiles per CU, so these cases are large: every (input channels, channel tile) shape of
# its launcher, stride 2 (a down-sampling branch), forward + statistics, data gradient, accumulate.
STREAM_CASES = [
    # name, N, Ci, Co, H, W, stride
    ("k64_n64", 64, 64, 64, 32, 32, 1),
    ("k64_n128", 64, 64, 128, 32, 32, 1),
    ("k64_n256_and_k256_n64", 64, 64, 256, 32, 32, 1),
    ("k128_n64", 256, 128, 64, 16, 16, 1),
    ("k128_n128", 256, 128, 128, 16, 16, 1),
    ("k128_n512", 128, 128, 512, 16, 16, 1),
    ("k64_n192", 64, 64, 192, 32, 32, 1),       # Co = 192: 64-wide channel tiles (only 64 / 128 / 256 are instantiated)
    ("k128_n192", 256, 128, 192, 16, 16, 1),
    ("k256_n128_and_k128_n256", 128, 256, 128, 16, 16, 1),
    ("k256_n512_s2", 64, 256, 512, 32, 32, 2),
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=[c[0] for c in STREAM_CASES])
def test_conv1x1_stream_kernel(case):
    name, n, ci, co, h, w, stride = case
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = bf16_round(torch.randn(n, ci, h, w, generator=g))
    wt = bf16_round(torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    xr = x.clone().requires_grad_(True)
    ref = F.conv2d(xr, wt, None, stride=stride)
    dz = bf16_round(torch.randn(ref.shape, generator=g))
    ref.backward(dz)
    xp = to_padded_nhwc(x, 1, 1, 1, 1)
    taps = tapset(1, 1, 1, 1, 1, 1, 0, 1, 1)
    y, stats = run_conv(xp, pack_fwd(wt), n, h + 2, w + 2, ci, ho, wo, 0, ho, wo, 1, 0, 0, stride, ci, co, taps, want_stats=True)
    got = from_nhwc(y, n, ho, wo, co, 0)
    assert rel_l2(got, ref.detach()) < REL_TOL
    s = stats.sum(dim=0).cpu()
    assert rel_l2(s[0], got.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_l2(s[1], (got * got).sum(dim=(0, 2, 3))) < 1e-4
    # padded output (the eval forward writes activations with their border): same values, border untouched
    yp, _ = run_conv(xp, pack_fwd(wt), n, h + 2, w + 2, ci, ho, wo, 1, ho, wo, 1, 0, 0, stride, ci, co, taps)
    ypv = yp.view(n, ho + 2, wo + 2, co)
    assert torch.equal(ypv[:, 1:-1, 1:-1], y.view(n, ho, wo, co))
    assert float(ypv[:, 0].abs().max()) == 0.0 and float(ypv[:, :, 0].abs().max()) == 0.0
    if stride != 1:
        return
    # data gradient (Co -> Ci), then once more on top (accumulate)
    dzp = to_padded_nhwc(dz, 1, 1, 1, 1)
    dx = torch.zeros(n * h * w * ci, dtype=torch.bfloat16, device="cuda")
    run_conv(dzp, pack_dgrad(wt), n, ho + 2, wo + 2, co, h, w, 0, h, w, 1, 0, 0, 1, co, ci, taps, y=dx)
    assert rel_l2(from_nhwc(dx, n, h, w, ci, 0), xr.grad) < REL_TOL
    run_conv(dzp, pack_dgrad(wt), n, ho + 2, wo + 2, co, h, w, 0, h, w, 1, 0, 0, 1, co, ci, taps, y=dx, accumulate=1)
    assert rel_l2(from_nhwc(dx, n, h, w, ci, 0), 2 * xr.grad) < 2 * REL_TOL


def _pack_mask_bits(keep):
    """bool [M][C] -> uint8 [M][C/8], bit j of byte (m, c8) = keep[m][8 c8 + j] (the fused forward BatchNorm's bit map)."""
    m, c = keep.shape
    k = keep.view(m, c // 8, 8).to(torch.int32)
    return (k * (2 ** torch.arange(8, dtype=torch.int32))).sum(dim=2).to(torch.uint8)


STREAM_EP_CASES = [
    # name, N, Ci, Co, H, W
    ("k64_n256", 32, 64, 256, 32, 32),       # a Bottleneck's closing 1x1 (eval: BatchNorm + identity + ReLU in the epilogue)
    ("k256_n64", 64, 256, 64, 32, 32),       # its opening 1x1; as data gradient 64 -> 256 it adds onto the masked identity path
    ("k128_n512", 128, 128, 512, 16, 16),
    ("k256_n128", 16, 256, 128, 32, 32),
]


@pytest.mark.parametrize("case", STREAM_EP_CASES, ids=[c[0] for c in STREAM_EP_CASES])
def test_conv1x1_stream_kernel_eval_epilogue_and_masked_accumulate(case):
    """vpd_op_conv2d_ep on shapes conv1x1_stream_kernel takes: (a) the eval epilogue relu(scale * conv + shift + residual) into a
    padded activation, against fp32 torch on the same bf16 operands; (b) y = old * mask + conv with the ReLU bit map, the way a
    Bottleneck's first data gradient lands on d(block output)."""
    name, n, ci, co, h, w = case
    L = _lib()
    g = torch.Generator().manual_seed(hash(name) % 1000 + 7)
    x = bf16_round(torch.randn(n, ci, h, w, generator=g))
    wt = bf16_round(torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5)
    conv = F.conv2d(x, wt)
    taps = tapset(1, 1, 1, 1, 1, 1, 0, 1, 1)
    xp = to_padded_nhwc(x, 1, 1, 1, 1)
    # (a) eval epilogue
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.3
    res = bf16_round(torch.randn(n, co, h, w, generator=g))
    for with_res in (False, True):
        y = torch.zeros(n * (h + 2) * (w + 2) * co, dtype=torch.bfloat16, device="cuda")
        resp = to_padded_nhwc(res, 1, 1, 1, 1) if with_res else None
        sc, sh = scale.cuda(), shift.cuda()
        _check(L.vpd_op_conv2d_ep(ptr(xp), ptr(pack_fwd(wt)), ptr(y), n, h + 2, w + 2, ci, h + 2, w + 2, co, 1, h, w, 1, ci, co, taps,
                                  ptr(sc), ptr(sh), ptr(resp) if with_res else None, 1, 0, None, stream()))
        torch.cuda.synchronize()
        ref = conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if with_res:
            ref = ref + res
        ref = ref.clamp_min(0)
        assert rel_l2(from_nhwc(y, n, h + 2, w + 2, co, 1), ref) < REL_TOL, with_res
        yv = y.view(n, h + 2, w + 2, co)
        assert float(yv[:, 0].abs().max()) == 0.0 and float(yv[:, :, -1].abs().max()) == 0.0
    # (b) masked accumulate onto a dense y
    old = bf16_round(torch.randn(n, co, h, w, generator=g))
    keep = torch.rand(n, h, w, co, generator=g) > 0.4
    yd = old.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda().flatten()
    bits = _pack_mask_bits(keep.view(-1, co)).cuda()
    _check(L.vpd_op_conv2d_ep(ptr(xp), ptr(pack_fwd(wt)), ptr(yd), n, h + 2, w + 2, ci, h, w, co, 0, h, w, 1, ci, co, taps,
                              None, None, None, 0, 1, ptr(bits), stream()))
    torch.cuda.synchronize()
    ref = conv + old * keep.permute(0, 3, 1, 2).float()
    assert rel_l2(from_nhwc(yd, n, h, w, co, 0), ref) < REL_TOL
