"""Host logic of the device input pipeline (row f1) and known answers of its CPU oracle -- no GPU needed."""
import ctypes as C
import math
import re

import numpy as np
import torch

from oracle import augment_oracle as AO
from vpd_amd import augment as A


def test_param_struct_layout_matches_header():
    """AUG_DTYPE (what the host uploads) is field for field the vpd_aug_params of include/vpd_hip.h."""
    import os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'vpd_hip.h')).read()
    body = re.search(r'typedef struct vpd_aug_params \{(.*?)\} vpd_aug_params;', hdr, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    decls = [d.strip() for d in body.split(';') if d.strip()]
    assert decls == ['int order[4]', 'float factor[4]', 'int flip', 'int noise', 'int crop_i, crop_j, crop_h, crop_w',
                     'unsigned int seed_lo, seed_hi']

    class S(C.Structure):
        _fields_ = [('order', C.c_int * 4), ('factor', C.c_float * 4), ('flip', C.c_int), ('noise', C.c_int),
                    ('crop', C.c_int * 4), ('seed', C.c_uint * 2)]
    assert C.sizeof(S) == A.AUG_DTYPE.itemsize == 64
    for f in ('order', 'factor', 'flip', 'noise', 'crop', 'seed'):
        assert getattr(S, f).offset == A.AUG_DTYPE.fields[f][1]


def test_sample_params_same_distribution_as_oracle():
    """The product sampler is vectorised (batched draws), the oracle restates torchvision's per-item get_params loops:
    different streams, same distributions -- compared on 4000 draws each (two-sample bounds far above sampling noise)."""
    n = 4000
    p = A.sample_params(n, 128, 128, generator=torch.Generator().manual_seed(11))
    g2 = torch.Generator().manual_seed(12)
    o = [AO.sample_item_params(128, 128, g2) for _ in range(n)]
    # ColorJitter: every one of the 24 op orders with frequency 1/24, factors uniform in their ranges
    perm_p = np.unique(p['order'], axis=0, return_counts=True)[1] / n
    perm_o = np.unique(np.array([x['order'] for x in o]), axis=0, return_counts=True)[1] / n
    assert len(perm_p) == 24 and len(perm_o) == 24
    assert np.abs(perm_p - 1 / 24).max() < 0.015 and np.abs(perm_o - 1 / 24).max() < 0.015
    fo = np.array([x['factors'] for x in o], dtype=np.float64)
    fp = p['factor'].astype(np.float64)
    for k, width in enumerate((0.4, 0.4, 0.1, 0.1)):
        assert abs(fp[:, k].mean() - fo[:, k].mean()) < 0.03 * width
        assert abs(fp[:, k].std() - width / np.sqrt(12)) < 0.03 * width and abs(fo[:, k].std() - width / np.sqrt(12)) < 0.03 * width
    assert abs(p['flip'].mean() - np.mean([x['flip'] for x in o])) < 0.04
    assert abs(p['noise'].mean() - np.mean([x['noise'] for x in o])) < 0.04
    # RandomResizedCrop windows: same mean / spread of height, width, area and of the offsets relative to their range
    co = np.array([x['crop'] for x in o], dtype=np.float64)
    cp = p['crop'].astype(np.float64)
    for k in (2, 3):
        assert abs(cp[:, k].mean() - co[:, k].mean()) < 0.6 and abs(cp[:, k].std() - co[:, k].std()) < 0.6
    assert abs((cp[:, 2] * cp[:, 3]).mean() - (co[:, 2] * co[:, 3]).mean()) < 0.01 * 128 * 128
    rel = lambda c: np.stack([c[:, 0] / np.maximum(128 - c[:, 2], 1), c[:, 1] / np.maximum(128 - c[:, 3], 1)], 1)
    m = (cp[:, 2] < 120) & (cp[:, 3] < 120)
    mo = (co[:, 2] < 120) & (co[:, 3] < 120)
    assert np.abs(rel(cp)[m].mean(0) - rel(co)[mo].mean(0)).max() < 0.03
    # a fresh noise key per call unless one is given (the same key every batch repeats the noise pattern)
    a = A.sample_params(4, 64, 64, generator=torch.Generator().manual_seed(1))
    g = torch.Generator().manual_seed(1)
    b1, b2 = A.sample_params(4, 64, 64, generator=g), A.sample_params(4, 64, 64, generator=g)
    assert tuple(a['seed'][0]) == tuple(b1['seed'][0]) and tuple(b1['seed'][0]) != tuple(b2['seed'][0])
    assert (b1['seed'] == b1['seed'][0]).all()
    assert tuple(A.sample_params(2, 64, 64, seed=(5 << 32) | 7)['seed'][1]) == (7, 5)


def test_sample_params_ranges():
    p = A.sample_params(500, 128, 128, generator=torch.Generator().manual_seed(3))
    assert sorted(set(map(tuple, np.sort(p['order'], axis=1)))) == [(0, 1, 2, 3)]
    f = p['factor']
    assert (f[:, 0] >= 0.8).all() and (f[:, 0] <= 1.2).all() and (f[:, 1] >= 0.8).all() and (f[:, 1] <= 1.2).all()
    assert (f[:, 2] >= 0.95).all() and (f[:, 2] <= 1.05).all() and (np.abs(f[:, 3]) <= 0.05).all()
    c = p['crop']
    area = c[:, 2] * c[:, 3] / (128.0 * 128.0)
    ratio = c[:, 3] / c[:, 2]
    assert (area > 0.48).all() and (area <= 1.0).all() and (ratio > 0.85).all() and (ratio < 1.16).all()
    assert (c[:, 0] + c[:, 2] <= 128).all() and (c[:, 1] + c[:, 3] <= 128).all()
    assert 0.35 < p['flip'].mean() < 0.65 and 0.35 < p['noise'].mean() < 0.65
    q = A.sample_params(7, 96, 128, augment=False)
    assert (q['order'] == -1).all() and (q['crop'] == (0, 0, 96, 128)).all() and not q['flip'].any()


def test_oracle_known_answers():
    g = torch.Generator().manual_seed(0)
    img = torch.rand(3, 16, 16, generator=g)
    # factor 1 / shift 0 are identities (hue round trip up to fp32 rounding)
    assert torch.equal(AO.adjust_brightness(img, 1.0), img)
    assert torch.allclose(AO.adjust_contrast(img, 1.0), img, atol=1e-7)
    assert torch.allclose(AO.adjust_saturation(img, 1.0), img, atol=1e-7)
    assert torch.allclose(AO.adjust_hue(img, 0.0), img, atol=2e-6)
    # saturation 0 = grey image in every channel; contrast 0 = the grey mean everywhere
    grey = AO.rgb_to_grayscale(img)
    assert torch.allclose(AO.adjust_saturation(img, 0.0), grey.expand(3, -1, -1).clamp(0, 1), atol=1e-7)
    assert torch.allclose(AO.adjust_contrast(img, 0.0), torch.full_like(img, float(grey.mean())), atol=1e-6)
    # hue: pure red shifted by 1/3 is pure green, by 1/2 cyan; a full turn is the identity
    red = torch.tensor([1.0, 0.0, 0.0]).view(3, 1, 1)
    assert torch.allclose(AO.adjust_hue(red, 1.0 / 3.0).flatten(), torch.tensor([0.0, 1.0, 0.0]), atol=1e-5)
    assert torch.allclose(AO.adjust_hue(red, 0.5).flatten(), torch.tensor([0.0, 1.0, 1.0]), atol=1e-5)
    assert torch.allclose(AO.adjust_hue(img, 1.0), img, atol=3e-6)


def test_oracle_identity_is_the_plain_loader():
    """No augmentation == what vpd_amd.data.load_rgb / load_flow (and reference common.py:52-69) produce."""
    rs = np.random.RandomState(5)
    rgb = rs.randint(0, 256, (24, 24, 3)).astype(np.uint8)
    flow = rs.randint(0, 256, (24, 24, 2)).astype(np.uint8)
    mean, std = (0.34, 0.46, 0.52), (0.16, 0.17, 0.19)
    p = AO.sample_item_params(24, 24, augment=False)
    out = AO.augment_item(rgb, flow, None, None, p, mean, std, 24)
    ref_rgb = (torch.from_numpy(rgb).float().permute(2, 0, 1) / 255. - torch.tensor(mean).view(3, 1, 1)) \
        / torch.tensor(std).view(3, 1, 1)
    ref_flow = torch.FloatTensor((flow / 255) - 0.5).permute(2, 0, 1)
    assert torch.equal(out[:3], ref_rgb) and torch.equal(out[3:], ref_flow)
    # flip: mirrored columns, x-flow negated, y-flow kept
    p['flip'] = True
    fl = AO.augment_item(rgb, flow, None, None, p, mean, std, 24)
    assert torch.equal(fl[:3], torch.flip(ref_rgb, (2,)))
    assert torch.equal(fl[3], -torch.flip(ref_flow[0], (1,))) and torch.equal(fl[4], torch.flip(ref_flow[1], (1,)))
    # mask noise lands only where the mask png is non-zero
    p['flip'], p['noise'] = False, True
    mask = (rs.rand(24, 24) > 0.5).astype(np.uint8) * 255
    noise = torch.randn(3, 24, 24, generator=torch.Generator().manual_seed(1))
    nz = AO.augment_item(rgb, flow, mask, noise, p, mean, std, 24)
    d = nz[:3] - ref_rgb
    m = torch.from_numpy(mask) != 0
    assert torch.all(d[:, ~m] == 0) and torch.allclose(d[:, m], noise[:, m] * math.sqrt(0.05), atol=1e-6)


def test_oracle_crop_resize_is_torch_bilinear():
    rs = np.random.RandomState(6)
    rgb = rs.randint(0, 256, (32, 32, 3)).astype(np.uint8)
    p = AO.sample_item_params(32, 32, augment=False)
    p['crop'] = (3, 5, 20, 22)
    mean, std = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    out = AO.augment_item(rgb, None, None, None, p, mean, std, 32)
    src = torch.from_numpy(rgb).float().permute(2, 0, 1)[:, 3:23, 5:27] / 255.
    # corners of a bilinear up-sampling with align_corners=False are the source corners
    assert out.shape == (3, 32, 32)
    assert torch.allclose(out[:, 0, 0], src[:, 0, 0]) and torch.allclose(out[:, -1, -1], src[:, -1, -1])
    assert float(out.min()) >= float(src.min()) - 1e-6 and float(out.max()) <= float(src.max()) + 1e-6
