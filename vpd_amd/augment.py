"""Train-time input pipeline on the GPU.

What the reference does per item on DataLoader workers (vpd_dataset/single_frame.py:168-206 with
vpd_dataset/common.py:49-92) -- u8 -> float, ColorJitter, Normalize, mask noise, flow decode, h-flip,
RandomResizedCrop -- runs here as two HIP launches per BATCH over raw u8 crops (5 B per pixel over PCIe
instead of 20 B of fp32).  The host keeps only the random *decisions*: ``sample_params`` draws them with
torchvision's algorithms (ColorJitter.get_params: randperm(4) then b, c, s, h; RandomResizedCrop.get_params: up to
ten (area, log-ratio) tries then randint i, j), i.e. from the same distributions.  The reference seeds nothing and
mixes Python's and torch's generators, so there is no reference random stream to reproduce.

    aug = CropAugmenter(device, RGB_MEAN_STD['diving48'], img_dim=128, use_flow=True)
    params = sample_params(n, 128, 128)                   # numpy structured array, one row per crop
    img = aug(rgb_u8, flow_u8, mask_u8, params)           # f32 [n, 5, 128, 128] on the device == batch['img']
    # or, fused with the stem's staging (no fp32 batch at all):
    aug.stage(engine, rgb_u8, flow_u8, mask_u8, params, train=True); engine.forward_train(None, tgt, staged=(n, 128))
"""
import ctypes as C
import math

import os

import numpy as np
import torch

from ._lib import check, lib

JITTER_KWARGS = {'brightness': 0.2, 'contrast': 0.2, 'saturation': 0.05, 'hue': 0.05}   # vpd_dataset/common.py:11-12
RANDOM_MASK_PROB = 0.5                        # vpd_dataset/single_frame.py:20
RANDOM_NOISE_SD = math.sqrt(0.05)             # vpd_dataset/single_frame.py:21
RRC_SCALE, RRC_RATIO = (0.5, 1.0), (0.9, 1.1)  # vpd_dataset/common.py:49-50

# binary layout of vpd_aug_params (include/vpd_hip.h), 64 bytes
AUG_DTYPE = np.dtype([('order', '<i4', (4,)), ('factor', '<f4', (4,)), ('flip', '<i4'), ('noise', '<i4'),
                      ('crop', '<i4', (4,)), ('seed', '<u4', (2,))])
assert AUG_DTYPE.itemsize == 64


def identity_params(n, height, width):
    """No augmentation: full window, no jitter, no flip (validation of Penn / inference views)."""
    p = np.zeros(n, dtype=AUG_DTYPE)
    p['order'] = -1
    p['factor'] = (1.0, 1.0, 1.0, 0.0)
    p['crop'] = (0, 0, height, width)
    return p


def sample_params(n, height, width, augment=True, flip=True, generator=None, seed=None):
    """Random decisions for n crops, drawn from torch's CPU RNG (`generator` or the global one) -- vectorised: a
    handful of batched draws per call instead of ~10 one-element draws per crop, so that the host side keeps up with
    the device (the per-crop loop capped the loop near 20 k crops/s).  Same distributions as torchvision's
    ColorJitter.get_params / RandomResizedCrop.get_params (up to ten (area, log-ratio) tries, first valid one wins,
    central fallback); the reference seeds nothing, so there is no stream to reproduce.

    seed: 64-bit Philox key of the device-side mask noise for this BATCH (the device counter is (pixel, batch slot)).
    None draws a fresh key from the same RNG: two batches never share a noise pattern."""
    p = identity_params(n, height, width)
    if not augment or n == 0:
        return p
    g = generator
    rand = lambda *shape: torch.rand(shape, generator=g, dtype=torch.float64).numpy()
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), generator=g, dtype=torch.int64))
    if flip:
        p['flip'] = torch.randint(0, 2, (n,), generator=g).numpy()
    # ColorJitter.get_params: a uniformly random permutation of the four ops (argsort of iid uniforms), then b, c, s, h
    p['order'] = np.argsort(rand(n, 4), axis=1).astype(np.int32)
    u = rand(n, 4)
    jk = JITTER_KWARGS
    lo = np.array([1 - jk['brightness'], 1 - jk['contrast'], 1 - jk['saturation'], -jk['hue']])
    hi = np.array([1 + jk['brightness'], 1 + jk['contrast'], 1 + jk['saturation'], jk['hue']])
    p['factor'] = (lo + u * (hi - lo)).astype(np.float32)
    p['noise'] = (rand(n) <= RANDOM_MASK_PROB).astype(np.int32)
    # RandomResizedCrop.get_params: ten candidate (area, aspect) pairs per crop, the first that fits is taken
    area = height * width
    ta = area * (RRC_SCALE[0] + rand(n, 10) * (RRC_SCALE[1] - RRC_SCALE[0]))
    lr0, lr1 = math.log(RRC_RATIO[0]), math.log(RRC_RATIO[1])
    aspect = np.exp(lr0 + rand(n, 10) * (lr1 - lr0))
    w = np.rint(np.sqrt(ta * aspect)).astype(np.int64)
    h = np.rint(np.sqrt(ta / aspect)).astype(np.int64)
    ok = (w > 0) & (w <= width) & (h > 0) & (h <= height)
    first = np.argmax(ok, axis=1)
    has = ok.any(axis=1)
    rows = np.arange(n)
    cw, ch = w[rows, first], h[rows, first]
    # fallback: central crop at the nearest allowed ratio
    in_ratio = float(width) / float(height)
    if in_ratio < min(RRC_RATIO):
        fw, fh = width, int(round(width / min(RRC_RATIO)))
    elif in_ratio > max(RRC_RATIO):
        fh, fw = height, int(round(height * max(RRC_RATIO)))
    else:
        fw, fh = width, height
    cw = np.where(has, cw, fw)
    ch = np.where(has, ch, fh)
    ui, uj = rand(n), rand(n)
    ci = np.where(has, np.minimum((ui * (height - ch + 1)).astype(np.int64), height - ch), (height - ch) // 2)   # randint(0, H-h+1)
    cj = np.where(has, np.minimum((uj * (width - cw + 1)).astype(np.int64), width - cw), (width - cw) // 2)
    p['crop'] = np.stack([ci, cj, ch, cw], axis=1).astype(np.int32)
    p['seed'] = (seed & 0xffffffff, (seed >> 32) & 0xffffffff)
    return p


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class CropAugmenter:
    """Device side of the pipeline for one (mean/std, img_dim, use_flow) configuration."""

    def __init__(self, device, rgb_mean_std, img_dim, use_flow, noise_sd=RANDOM_NOISE_SD):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('CropAugmenter runs HIP kernels: it needs a GPU device (no CPU fallback)')
        self.mean_std6 = [float(v) for v in rgb_mean_std[0]] + [float(v) for v in rgb_mean_std[1]]
        self.img_dim, self.use_flow, self.noise_sd = img_dim, use_flow, float(noise_sd)
        self._scratch = None

    def _check(self, rgb_u8, flow_u8, mask_u8, params, noise):
        assert rgb_u8.dtype == torch.uint8 and rgb_u8.dim() == 4 and rgb_u8.shape[3] == 3 and rgb_u8.is_contiguous() \
            and rgb_u8.is_cuda, 'rgb_u8 must be a contiguous device u8 [N,H,W,3]'
        n, h, w, _ = rgb_u8.shape
        if self.use_flow:
            assert flow_u8 is not None and flow_u8.dtype == torch.uint8 and tuple(flow_u8.shape) == (n, h, w, 2) \
                and flow_u8.is_contiguous() and flow_u8.is_cuda, 'flow_u8 must be a device u8 [N,H,W,2]'
        else:
            assert flow_u8 is None, 'flow_u8 given to a 3-channel pipeline'
        if mask_u8 is not None:
            assert mask_u8.dtype == torch.uint8 and tuple(mask_u8.shape) == (n, h, w) and mask_u8.is_contiguous() \
                and mask_u8.is_cuda
        if noise is not None:
            assert noise.dtype == torch.float32 and tuple(noise.shape) == (n, 3, h, w) and noise.is_contiguous() \
                and noise.is_cuda
        assert params.dtype == AUG_DTYPE and params.shape == (n,), 'params: one vpd_aug_params row per crop'
        c = params['crop']
        assert (c[:, 0] >= 0).all() and (c[:, 1] >= 0).all() and (c[:, 2] >= 1).all() and (c[:, 3] >= 1).all() and \
            (c[:, 0] + c[:, 2] <= h).all() and (c[:, 1] + c[:, 3] <= w).all(), 'crop window outside the image'
        if self._scratch is None or self._scratch.numel() < 8 * n:       # 8 partial grey sums per crop
            self._scratch = torch.empty(8 * max(n, 256), dtype=torch.float32, device=self.device)
        pdev = torch.from_numpy(params.view(np.uint8).reshape(n, 64)).to(self.device, non_blocking=True)
        return n, h, w, pdev

    def __call__(self, rgb_u8, flow_u8, mask_u8, params, noise=None, out=None):
        n, h, w, pdev = self._check(rgb_u8, flow_u8, mask_u8, params, noise)
        c = 5 if self.use_flow else 3
        if out is None:
            out = torch.empty((n, c, self.img_dim, self.img_dim), dtype=torch.float32, device=self.device)
        ms = (C.c_float * 6)(*self.mean_std6)
        check(lib().vpd_augment_crops(_ptr(rgb_u8), _ptr(flow_u8), _ptr(mask_u8), _ptr(noise), _ptr(pdev), n, h, w,
                                      self.img_dim, ms, self.noise_sd, _ptr(out), _ptr(self._scratch),
                                      C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
              'vpd_augment_crops')
        return out

    def stage(self, engine, rgb_u8, flow_u8, mask_u8, params, train, motion=False, noise=None):
        """Write the augmented batch straight into `engine`'s stem staging buffer (bf16 NHWC)."""
        n, h, w, pdev = self._check(rgb_u8, flow_u8, mask_u8, params, noise)
        engine.stage_crops(rgb_u8, flow_u8, mask_u8, pdev, noise, self.img_dim, self.mean_std6, self.noise_sd,
                           self._scratch, train, motion)
        return n, self.img_dim

    def stage_views(self, engine, rgb_u8, flow_u8, flip):
        """Inference views of FrameDataset (vpd_dataset/single_frame.py:377-400) for n decoded frames, written straight into
        the EVAL plan's stem staging buffer: k = 2 views per frame in the order [orig, h-flip] (x-flow negated in the flipped
        view) when `flip`, else the frame itself.  Returns (n * k, img_dim) for forward_eval / the staged eval graph."""
        n, h, w, _ = rgb_u8.shape
        k = 2 if flip else 1
        if h == self.img_dim and w == self.img_dim and w % 4 == 0 and os.environ.get("VPD_FAST_VIEWS", "1") != "0":
            # frames already have the model's size (the apply path): the dedicated table-driven kernel, no parameter
            # records, no duplicated frames (VPD_FAST_VIEWS=0: the general pipeline with identity parameters, below)
            engine.stage_views(rgb_u8, flow_u8, k, self.mean_std6)
            return n * k, self.img_dim
        key = (n, k, h, w)
        cache = self.__dict__.setdefault('_view_params', {})
        if key not in cache:
            p = identity_params(n * k, h, w)
            if flip:
                p['flip'][1::2] = 1
            cache[key] = (p, torch.from_numpy(p.view(np.uint8).reshape(n * k, 64)).to(self.device))
        p, pdev = cache[key]
        if k == 2:
            rgb_u8 = rgb_u8.repeat_interleave(2, dim=0)
            flow_u8 = flow_u8.repeat_interleave(2, dim=0) if flow_u8 is not None else None
        if self._scratch is None or self._scratch.numel() < 8 * n * k:
            self._scratch = torch.empty(8 * max(n * k, 256), dtype=torch.float32, device=self.device)
        engine.stage_crops(rgb_u8, flow_u8, None, pdev, None, self.img_dim, self.mean_std6, self.noise_sd, self._scratch,
                           False, False)
        return n * k, self.img_dim
