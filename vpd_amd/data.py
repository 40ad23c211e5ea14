"""Batch producers on either side of the hot path (the
*contract* is in scope, the augmentation pipeline is a "next" row).

* RGB_MEAN_STD, EMB_FILE_SUFFIX .... reference vpd_dataset/common.py:9-36
* TeacherEmbDataset ................ GenericDataset (vpd_dataset/single_frame.py:166-273):
      teacher .emb.pkl ingestion (pose-score filter :35-44/:236-242, embed_time pairing
      :247-258), unseeded 80/20 split (:263), sampling WITH replacement for a constant
      epoch length (common.py:104-108), h-flip with x-flow negation (:199-203).
      ColorJitter, mask noise and RandomResizedCrop run on the DEVICE (vpd_amd/augment.py): with raw_u8=True
      an item is the decoded u8 arrays + the flip decision, and ModelTrainer(augmenter=...) does the rest.
      The CPU float path (raw_u8=False) does flips only.
* FrameDataset ..................... vpd_dataset/single_frame.py:361-403 (orig + h-flip views)
* SyntheticCrops ................... seeded crops in the reference's value ranges

PNG decoding uses PIL (cv2 is not in this image).
"""
import math
import os
import random
import re

import numpy as np
import torch

from .io import load_pickle

EMB_FILE_SUFFIX = '.emb.pkl'
DEFAULT_MIN_POSE_SCORE = 0.5

RGB_MEAN_STD = {
    'tennis': ((0.44157383614877077, 0.47029633580897046, 0.4534017568516162),
               (0.13526736314774856, 0.1208027074415591, 0.1261687563723076)),
    'fs': ((0.5747710337842444, 0.5644043210903272, 0.6334494151377134),
           (0.21349823115367886, 0.21827191146692457, 0.20393919008463163)),
    'fx': ((0.38402001736617936, 0.34764328219285123, 0.4099846773620623),
           (0.19505844565544309, 0.18984186888162677, 0.1989230425908947)),
    'diving48': ((0.3411329922282787, 0.46349889258964044, 0.5162481674015696),
                 (0.16302619019820488, 0.17092395707914718, 0.19266662199338647)),
    'penn': ((0.43258389316320306, 0.4293850246457961, 0.383481774195889),
             (0.18936336742486998, 0.18502009571154798, 0.18244625387985822)),
    'resnet': ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
}


def _load_png(path, img_dim):
    from PIL import Image
    im = Image.open(path).convert('RGB')
    if im.size != (img_dim, img_dim):
        im = im.resize((img_dim, img_dim), Image.BILINEAR)
    return np.asarray(im, dtype=np.float32)          # H, W, 3 (RGB)


def load_rgb_u8(path, img_dim):
    """The decoded crop as it is stored: u8 [H, W, 3] (RGB).  Normalisation happens on the device (vpd_amd.augment)."""
    from PIL import Image
    im = Image.open(path).convert('RGB')
    if im.size != (img_dim, img_dim):
        im = im.resize((img_dim, img_dim), Image.BILINEAR)
    return np.asarray(im, dtype=np.uint8)


def load_flow_u8(path, img_dim):
    """u8 [H, W, 2]: the two channels load_flow keeps (cv2's BGR channels 0 and 1 of the RGB-ordered PNG), undecoded."""
    return np.ascontiguousarray(load_rgb_u8(path, img_dim)[:, :, ::-1][:, :, :2])


def load_rgb(path, img_dim, rgb_mean_std):
    rgb = torch.from_numpy(_load_png(path, img_dim)).permute(2, 0, 1) / 255.
    mean = torch.tensor(rgb_mean_std[0]).view(3, 1, 1)
    std = torch.tensor(rgb_mean_std[1]).view(3, 1, 1)
    return (rgb - mean) / std


def load_flow(path, img_dim):
    # raft/flow.py:80-84 stores (fx, fy, 128) as RGB-ordered PNG channels read back by cv2 as BGR;
    # vpd_dataset/common.py:69 keeps cv2 channels 0 and 1
    # (u8 / 255 - 0.5 in float64, rounded to fp32 once: numpy's arithmetic on the reference's uint8 array, common.py:69)
    arr = _load_png(path, img_dim)[:, :, ::-1][:, :, :2].astype(np.float64)
    return torch.from_numpy(arr / 255. - 0.5).permute(2, 0, 1).float()


def _pose_score(meta):
    """_get_pose_score (vpd_dataset/single_frame.py:35-44): dp_score, else kp_score, else NotImplementedError -- a
    teacher pickle without either score cannot be filtered (student pickles carry {} and are not training targets)."""
    for k in ('dp_score', 'kp_score'):
        if meta.get(k) is not None:
            return meta[k]
    raise NotImplementedError('teacher embedding meta has neither dp_score nor kp_score')


class TeacherEmbDataset(torch.utils.data.Dataset):

    def __init__(self, data, img_dir, img_dim, rgb_mean_std, target_len, flow_img_name=None, augment=True,
                 raw_u8=False):
        self.data, self.img_dir, self.img_dim = data, img_dir, img_dim
        self.rgb_mean_std, self.target_len = rgb_mean_std, target_len
        self.flow_img_name, self.augment, self.raw_u8 = flow_img_name, augment, raw_u8

    def _raw_item(self, video_name, frame_num, emb, flip):
        """Decoded u8 arrays for the device pipeline: rgb [H,W,3], flow [H,W,2] (x, y), mask [H,W] (all zero
        -> no noise anywhere -- when the item has no <frame>.mask.png, single_frame.py:181-183)."""
        d = os.path.join(self.img_dir, video_name)
        item = {'emb': torch.as_tensor(emb, dtype=torch.float32), 'flip': int(flip),
                'rgb_u8': torch.from_numpy(_load_png(os.path.join(d, '{}.png'.format(frame_num)),
                                                     self.img_dim).astype(np.uint8))}
        if self.flow_img_name is not None:
            fl = _load_png(os.path.join(d, '{}.{}.png'.format(frame_num, self.flow_img_name)), self.img_dim)
            item['flow_u8'] = torch.from_numpy(fl[:, :, ::-1][:, :, :2].astype(np.uint8).copy())
        mask_path = os.path.join(d, '{}.mask.png'.format(frame_num))
        if os.path.exists(mask_path):
            item['mask_u8'] = torch.from_numpy(_load_png(mask_path, self.img_dim)[:, :, 2].astype(np.uint8).copy())
        else:
            item['mask_u8'] = torch.zeros((self.img_dim, self.img_dim), dtype=torch.uint8)
        return item

    def __len__(self):
        return self.target_len

    def __getitem__(self, idx):
        video_name, frame_num, emb, _ = random.choice(self.data)      # index ignored, as the reference
        flip = False
        if len(emb.shape) == 2:
            flip = self.augment and random.getrandbits(1) > 0
            emb = emb[int(flip), :]
        if self.raw_u8:
            return self._raw_item(video_name, frame_num, emb, flip)
        img = load_rgb(os.path.join(self.img_dir, video_name, '{}.png'.format(frame_num)), self.img_dim,
                       self.rgb_mean_std)
        if self.flow_img_name is not None:
            flow = load_flow(os.path.join(self.img_dir, video_name, '{}.{}.png'.format(frame_num, self.flow_img_name)),
                             self.img_dim)
            img = torch.cat((img, flow))
        if flip:
            img = torch.flip(img, (2,))
            if self.flow_img_name is not None:
                img[3, :, :] *= -1
        return {'emb': torch.as_tensor(emb, dtype=torch.float32), 'img': img}

    @staticmethod
    def load_default(emb_dir, img_dir, img_dim, embed_time, target_len, rgb_mean_std, flow_img_name=None,
                     min_pose_score=None, exclude_prefixes=None, split_seed=None):
        """split_seed: None = unseeded 80/20 split like the reference (vpd_dataset/single_frame.py:263); data-parallel
        runs pass one seed to every rank so that all replicas train and validate on the same frames."""
        all_data, emb_dim = [], None
        thresh = DEFAULT_MIN_POSE_SCORE if min_pose_score is None else min_pose_score
        for emb_file in sorted(os.listdir(emb_dir)):
            if not emb_file.endswith(EMB_FILE_SUFFIX):
                continue
            video_name = emb_file.split(EMB_FILE_SUFFIX)[0]
            if exclude_prefixes is not None and video_name.startswith(tuple(exclude_prefixes)):
                print('Excluded:', video_name)
                continue
            video_embs = load_pickle(os.path.join(emb_dir, emb_file))
            for i, (frame_num, emb_target, emb_meta) in enumerate(video_embs):
                if emb_dim is None:
                    emb_dim = emb_target.shape[-1]
                assert emb_target.shape[-1] == emb_dim, 'Inconsistent emb dims {} != {}'.format(
                    emb_target.shape[-1], emb_dim)
                if _pose_score(emb_meta) < thresh:
                    continue
                if embed_time:
                    if i == 0 or video_embs[i - 1][0] != frame_num - 1:
                        continue
                    emb_prev = video_embs[i - 1][1]
                    emb_target = np.concatenate([emb_target, emb_target - emb_prev],
                                                axis=0 if len(emb_target.shape) == 1 else 1)
                all_data.append((video_name, frame_num, emb_target, emb_meta))
        print('Videos:', len({x[0] for x in all_data}))
        all_data.sort(key=lambda x: x[:2])            # listing order must not matter for a seeded split
        (random if split_seed is None else random.Random(split_seed)).shuffle(all_data)      # train_test_split(test_size=0.2)
        n_val = int(math.ceil(0.2 * len(all_data)))   # sklearn: n_test = ceil(test_size * n)
        val_data, train_data = sorted(all_data[:n_val], key=lambda x: x[:2]), sorted(all_data[n_val:], key=lambda x: x[:2])
        mk = lambda d, n: TeacherEmbDataset(d, img_dir, img_dim, rgb_mean_std, n, flow_img_name=flow_img_name)
        return mk(train_data, target_len), mk(val_data, int(target_len * 0.2)), emb_dim


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)


def _grey(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(dim=-3)


def _adjust_hue(img, f):
    """torchvision's adjust_hue for float CHW tensors (RGB -> HSV, h = (h + f) mod 1, HSV -> RGB)."""
    r, g, b = img.unbind(dim=-3)
    maxc, minc = torch.max(img, dim=-3).values, torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    div = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
    h = (maxc == r) * (bc - gc) + ((maxc == g) & (maxc != r)) * (2.0 + rc - bc) + \
        ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = (torch.fmod(h / 6.0 + 1.0, 1.0) + f) % 1.0
    v = maxc
    i = torch.floor(h * 6.0)
    fr = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - s * fr), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - s * (1.0 - fr)), 0.0, 1.0)
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a4 = torch.stack((torch.stack((v, q, p, p, t, v), dim=-3), torch.stack((t, v, v, q, p, p), dim=-3),
                      torch.stack((p, p, t, v, v, q), dim=-3)), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(img.dtype), a4)


def color_jitter(img, brightness=0.2, contrast=0.2, saturation=0.05, hue=0.05):
    """transforms.ColorJitter(**JITTER_KWARGS) (vpd_dataset/common.py:11-12) on a float CHW tensor: the four ops in a
    random order with factors drawn as ColorJitter.get_params does.  The inference views of the reference apply it to
    the NORMALISED image (vpd_dataset/single_frame.py:366-379), whose values outside [0, 1] the
    ops clamp -- reproduced as is."""
    order = torch.randperm(4).tolist()
    u = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi))
    fb, fc, fs, fh = u(1 - brightness, 1 + brightness), u(1 - contrast, 1 + contrast), \
        u(1 - saturation, 1 + saturation), u(-hue, hue)
    for op in order:
        if op == 0:
            img = _blend(img, torch.zeros_like(img), fb)
        elif op == 1:
            img = _blend(img, torch.mean(_grey(img), dim=(-3, -2, -1), keepdim=True), fc)
        elif op == 2:
            img = _blend(img, _grey(img), fs)
        else:
            img = _adjust_hue(img, fh)
    return img


def load_tennis_default(emb_dir, img_dir, img_dim, embed_time, target_len, rgb_mean_std, flow_img_name=None,
                        min_pose_score=None, exclude_prefixes=None, split_seed=None):
    """TennisDataset.load_default (vpd_dataset/single_frame.py:88-162): teacher pickles are named
    <player>__<video>_<start>_<end>.emb.pkl with clip-relative frame numbers, crops live in
    <img_dir>/<video>/<player>/<start + frame>.png, and the 80/20 split is over CLIPS, not frames."""
    thresh = DEFAULT_MIN_POSE_SCORE if min_pose_score is None else min_pose_score
    clips, emb_dim = [], None
    for emb_file in sorted(os.listdir(emb_dir)):
        if not emb_file.endswith(EMB_FILE_SUFFIX):
            continue
        name = emb_file.split(EMB_FILE_SUFFIX)[0]
        if exclude_prefixes is not None and name.startswith(tuple(exclude_prefixes)):
            print('Excluded:', name)
            continue
        video_embs = load_pickle(os.path.join(emb_dir, emb_file))
        clips.append((name, video_embs))
        if emb_dim is None:
            emb_dim = video_embs[0][1].shape[-1]
        assert emb_dim == video_embs[0][1].shape[-1]

    def items(clip_list):
        out = []
        for name, video_embs in clip_list:
            player, rest = name.split('__', 1)
            video_name, start_frame, _ = rest.rsplit('_', 2)
            for i, (frame_num, emb_target, emb_meta) in enumerate(video_embs):
                if _pose_score(emb_meta) < thresh:
                    continue
                if embed_time:
                    if i == 0 or video_embs[i - 1][0] != frame_num - 1:
                        continue
                    emb_target = np.concatenate([emb_target, emb_target - video_embs[i - 1][1]],
                                                axis=0 if len(emb_target.shape) == 1 else 1)
                out.append((os.path.join(video_name, player), int(start_frame) + frame_num, emb_target, emb_meta))
        return out

    print('Videos:', len(clips))
    (random if split_seed is None else random.Random(split_seed)).shuffle(clips)      # train_test_split(videos, test_size=0.2)
    n_val = int(math.ceil(0.2 * len(clips)))         # sklearn: n_test = ceil(test_size * n)
    val, train = items(clips[:n_val]), items(clips[n_val:])
    key = lambda x: x[:2]
    mk = lambda d, n: TeacherEmbDataset(sorted(d, key=key), img_dir, img_dim, rgb_mean_std, n, flow_img_name=flow_img_name)
    return mk(train, target_len), mk(val, int(target_len * 0.2)), emb_dim


class FrameDataset(torch.utils.data.Dataset):
    """Inference items {'video', 'frame', 'img': [k, C, H, W]}: views [orig, j x jitter(orig), j x jitter(flip), flip]
    (reference vpd_dataset/single_frame.py:361-403, order of Appendix B.8)."""

    def __init__(self, tasks, img_dim, rgb_mean_std, augment_jitter=0, augment_flip=False, flow_img_name=None, raw_u8=False):
        """raw_u8: items are {'video', 'frame', 'rgb_u8': u8[H,W,3], 'flow_u8': u8[H,W,2]} -- the decoded PNGs as stored,
        82 KB per frame instead of 655 KB of fp32 views; vpd_amd.apply.embed_dataset builds the views [orig, flip] on the
        device (normalise, flow decode, h-flip with x-flow negation).  Jittered views need the host path (ColorJitter on
        the NORMALISED image, Appendix B.8)."""
        self.jitter_count = int(augment_jitter or 0)
        self.tasks, self.img_dim, self.rgb_mean_std = tasks, img_dim, rgb_mean_std
        self.flip, self.flow_img_name = augment_flip, flow_img_name
        self.raw_u8 = bool(raw_u8)
        if self.raw_u8 and self.jitter_count:
            raise ValueError('raw_u8 items carry no jittered views: use the fp32 path with --jitter')

    def __len__(self):
        return len(self.tasks)

    def __getitem__(self, idx):
        video, frame_num, prefix = self.tasks[idx]
        if self.raw_u8:
            item = {'video': video, 'frame': frame_num,
                    'rgb_u8': torch.from_numpy(load_rgb_u8('{}.png'.format(prefix), self.img_dim))}
            if self.flow_img_name is not None:
                item['flow_u8'] = torch.from_numpy(load_flow_u8('{}.{}.png'.format(prefix, self.flow_img_name), self.img_dim))
            return item
        img = load_rgb('{}.png'.format(prefix), self.img_dim, self.rgb_mean_std)
        imgs = [img] + [color_jitter(img) for _ in range(self.jitter_count)]
        flips = []
        if self.flip:
            flip_img = torch.flip(img, (2,))
            flips = [flip_img]
            imgs += [color_jitter(flip_img) for _ in range(self.jitter_count)]      # jittered flips join `imgs` (B.8)
        if self.flow_img_name is not None:
            flow = load_flow('{}.{}.png'.format(prefix, self.flow_img_name), self.img_dim)
            imgs = [torch.cat((x, flow)) for x in imgs]
            if flips:
                ff = torch.flip(flow, (2,))
                ff[0, :, :] *= -1
                flips = [torch.cat((x, ff)) for x in flips]
        return {'video': video, 'frame': frame_num, 'img': torch.stack(imgs + flips)}


def list_crop_dir(crop_dir):
    """apply_vpd_model.get_dataset (:69-89): every <crop_dir>/<video>/<frame>.png"""
    img_re = re.compile(r'^\d+\.png$')
    tasks, videos = [], []
    for video_name in sorted(os.listdir(crop_dir)):
        d = os.path.join(crop_dir, video_name)
        if not os.path.isdir(d):
            continue
        vid = len(videos)
        videos.append(video_name)
        for f in os.listdir(d):
            if img_re.match(f):
                fr = int(os.path.splitext(f)[0])
                tasks.append((vid, fr, os.path.join(d, str(fr))))
    return videos, tasks


def list_tennis_crops(video_dir, crop_dir):
    """apply_vpd_model.get_tennis_dataset (:36-66): every <name>_<start>_<end>.mp4 yields two "videos",
    front__<clip> and back__<clip>, whose crops live in <crop_dir>/<name>/<player>/<absolute frame>.png; the frame
    number stored in the pickle is relative to the clip's start."""
    tasks, videos = [], []
    for video_file in sorted(os.listdir(video_dir)):
        if not video_file.endswith('.mp4'):
            continue
        clip = os.path.splitext(video_file)[0]
        src_name, start, end = clip.rsplit('_', 2)
        start, end = int(start), int(end)
        for player in ('front', 'back'):
            vid = len(videos)
            videos.append('{}__{}'.format(player, clip))
            count = 0
            for fr in range(start, end + 1):
                prefix = os.path.join(crop_dir, src_name, player, str(fr))
                if os.path.isfile(prefix + '.png'):
                    tasks.append((vid, fr - start, prefix))
                    count += 1
            if count == 0:
                print('{} has no crops'.format(videos[-1]))
    return videos, tasks


class SyntheticCrops(torch.utils.data.Dataset):
    """Seeded synthetic crops + teacher targets in the reference's value ranges (no files)."""

    def __init__(self, length, c_in, img_dim, emb_dim, motion, rgb_mean_std, seed=0, raw_u8=False):
        self.raw_u8 = raw_u8
        self.length, self.c_in, self.img_dim, self.emb_dim, self.motion = length, c_in, img_dim, emb_dim, motion
        self.mean = torch.tensor(rgb_mean_std[0]).view(3, 1, 1)
        self.std = torch.tensor(rgb_mean_std[1]).view(3, 1, 1)
        self.seed = seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        rgb = torch.randint(0, 256, (3, self.img_dim, self.img_dim), generator=g)
        fl = None
        if self.c_in > 3:
            fl = (124 + 12 * torch.randn((self.c_in - 3, self.img_dim, self.img_dim), generator=g)).round().clamp(0, 255)
        t = torch.randn(self.emb_dim, generator=g)
        if self.motion:
            t = torch.cat((t, t - torch.randn(self.emb_dim, generator=g)))
        if self.raw_u8:      # the same crops as decoded u8 arrays, for the device input pipeline
            item = {'emb': t, 'rgb_u8': rgb.permute(1, 2, 0).to(torch.uint8).contiguous(), 'flip': 0,
                    'mask_u8': torch.full((self.img_dim, self.img_dim), 255, dtype=torch.uint8)}
            if fl is not None:
                item['flow_u8'] = fl.permute(1, 2, 0).to(torch.uint8).contiguous()
            return item
        img = (rgb.float() / 255. - self.mean) / self.std
        if fl is not None:
            img = torch.cat((img, fl / 255. - 0.5))
        return {'emb': t, 'img': img}
