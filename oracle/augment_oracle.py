"""CPU oracle of the train-time input pipeline -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file; nothing under vpd_amd/ does.  It restates, in plain fp32 torch-CPU,
what one item of the reference's GenericDataset goes through between the decoded PNGs and batch['img']:

  reference vpd_dataset/common.py:52-60   rgb u8 HWC -> float / 255 -> transform
  reference vpd_dataset/common.py:87-92   transform = Compose([ColorJitter(**JITTER_KWARGS), Normalize(mean, std)])
  reference vpd_dataset/common.py:11-12   JITTER_KWARGS brightness .2 contrast .2 saturation .05 hue .05
  reference vpd_dataset/single_frame.py:178-191  mask noise: randn * sqrt(0.05), zeroed where mask png == 0
  reference vpd_dataset/common.py:62-69   flow: (u8[:, :, :2] / 255) - 0.5 (float64, then FloatTensor)
  reference vpd_dataset/single_frame.py:193-203  cat, h-flip, negate channel 3
  reference vpd_dataset/common.py:49-50, :80  RandomResizedCrop(img_dim, scale=(.5, 1), ratio=(.9, 1.1))

PARITY UNPINNED for the torchvision pieces: torchvision (requirements.txt:2, version not pinned) is absent from this
image and from /root/reference, and the reference holds no test vectors for its data pipeline.  ColorJitter /
Normalize / RandomResizedCrop below restate torchvision's published tensor algorithms (transforms/_functional_tensor.py:
_blend, rgb_to_grayscale weights 0.2989/0.587/0.114, _rgb2hsv, _hsv2rgb; transforms.py: ColorJitter.get_params /
forward op order, RandomResizedCrop.get_params); the resize is torch.nn.functional.interpolate(mode='bilinear',
align_corners=False), the routine torchvision's resized_crop calls (for the scale range used here every resize is
an up-sampling, where its antialias flag has no effect beyond rounding).
"""
import math

import torch
import torch.nn.functional as F

JITTER_KWARGS = {'brightness': 0.2, 'contrast': 0.2, 'saturation': 0.05, 'hue': 0.05}
RANDOM_MASK_PROB = 0.5
RANDOM_NOISE_SD = math.sqrt(0.05)
RRC_SCALE = (0.5, 1.0)
RRC_RATIO = (0.9, 1.1)


# ---- torchvision.transforms._functional_tensor (float images in [0, 1], CHW) ----
def _blend(img1, img2, ratio):
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 1.0)


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(dim=-3)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img, f):
    mean = torch.mean(rgb_to_grayscale(img), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def adjust_saturation(img, f):
    return _blend(img, rgb_to_grayscale(img), f)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = hr + hg + hb
    h = torch.fmod((h / 6.0 + 1.0), 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp((v * (1.0 - s)), 0.0, 1.0)
    q = torch.clamp((v * (1.0 - s * f)), 0.0, 1.0)
    t = torch.clamp((v * (1.0 - (s * (1.0 - f)))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img, f):
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(dim=-3)
    h = (h + f) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


# ---- random draws, in torchvision's order of RNG calls (global torch RNG, or `g`) ----
def color_jitter_params(g=None):
    """ColorJitter.get_params: randperm(4), then one uniform per enabled op (b, c, s, h)."""
    order = torch.randperm(4, generator=g).tolist()
    u = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi, generator=g))
    b = u(1 - JITTER_KWARGS['brightness'], 1 + JITTER_KWARGS['brightness'])
    c = u(1 - JITTER_KWARGS['contrast'], 1 + JITTER_KWARGS['contrast'])
    s = u(1 - JITTER_KWARGS['saturation'], 1 + JITTER_KWARGS['saturation'])
    h = u(-JITTER_KWARGS['hue'], JITTER_KWARGS['hue'])
    return order, (b, c, s, h)


def random_resized_crop_params(height, width, g=None):
    """RandomResizedCrop.get_params(img, scale=(.5, 1), ratio=(.9, 1.1)) -> (i, j, h, w)."""
    area = height * width
    log_ratio = (math.log(RRC_RATIO[0]), math.log(RRC_RATIO[1]))
    for _ in range(10):
        target_area = area * float(torch.empty(1).uniform_(RRC_SCALE[0], RRC_SCALE[1], generator=g))
        aspect = math.exp(float(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=g)))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, size=(1,), generator=g))
            j = int(torch.randint(0, width - w + 1, size=(1,), generator=g))
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(RRC_RATIO):
        w = width
        h = int(round(w / min(RRC_RATIO)))
    elif in_ratio > max(RRC_RATIO):
        h = height
        w = int(round(h * max(RRC_RATIO)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def sample_item_params(height, width, g=None, augment=True):
    """All random decisions of one GenericDataset item, as a dict (see vpd_amd.augment.AugParams)."""
    p = {'order': [-1, -1, -1, -1], 'factors': (1.0, 1.0, 1.0, 0.0), 'flip': False, 'noise': False,
         'crop': (0, 0, height, width)}
    if not augment:
        return p
    p['flip'] = bool(torch.randint(0, 2, (1,), generator=g))
    p['order'], p['factors'] = color_jitter_params(g)
    p['noise'] = float(torch.rand(1, generator=g)) <= RANDOM_MASK_PROB
    p['crop'] = random_resized_crop_params(height, width, g)
    return p


def augment_item(rgb_u8, flow_u8, mask_u8, noise, p, mean, std, out_dim):
    """rgb_u8 [H,W,3] u8 (RGB), flow_u8 [H,W,2] u8 or None, mask_u8 [H,W] u8 or None (noise is zeroed where it is 0),
    noise f32 [3,H,W] (standard normal draws) or None, p = sample_item_params(...) -> f32 [C, out_dim, out_dim]."""
    img = torch.as_tensor(rgb_u8).float().permute(2, 0, 1) / 255.
    for op in p['order']:                       # ColorJitter.forward: ops in the sampled order
        if op >= 0:
            img = _OPS[op](img, p['factors'][op])
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    img = (img - m) / s                         # transforms.Normalize
    if p['noise'] and noise is not None and mask_u8 is not None:
        nz = noise.clone().float() * RANDOM_NOISE_SD
        nz[:, torch.as_tensor(mask_u8) == 0] = 0
        img = img + nz
    if flow_u8 is not None:
        fl = torch.as_tensor((torch.as_tensor(flow_u8).double() / 255) - 0.5).float().permute(2, 0, 1)
        img = torch.cat((img, fl))
    if p['flip']:
        img = torch.flip(img, (2,))
        if flow_u8 is not None:
            img[3, :, :] *= -1
    i, j, h, w = p['crop']
    if (i, j, h, w) != (0, 0, img.shape[1], img.shape[2]) or out_dim != img.shape[1]:
        img = img[:, i:i + h, j:j + w]
        img = F.interpolate(img.unsqueeze(0), size=(out_dim, out_dim), mode='bilinear', align_corners=False)[0]
    return img.contiguous()
