#!/usr/bin/env python3
"""Drop-in for reference apply_vpd_model.py (same flags and output files)."""
import argparse
import os

import torch
from torch.utils.data import DataLoader

from vpd_amd import paths as dataset_paths
from vpd_amd.apply import StreamingWriter, apply_batch_size, embed_dataset
from vpd_amd.data import FrameDataset, list_crop_dir, list_tennis_crops
from vpd_amd.io import load_json
from vpd_amd.models.rgb import RGBF_EmbeddingModel


def get_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('model_dir', type=str)
    parser.add_argument('-d', '--dataset', type=str, required=True, choices=['tennis', 'fs', 'fx', 'diving48'])
    parser.add_argument('-o', '--out_dir', type=str)
    parser.add_argument('-m', '--model_epoch', type=int, help='Specify an epooh. Otherwise use the best one.')
    parser.add_argument('--jitter', type=int, help='Create additional jittered features.')
    parser.add_argument('--no_flip', action='store_true', help='Do not embed horizontal flips')
    parser.add_argument('--flow_img', type=str)
    parser.add_argument('--host_fp32', action='store_true',
                        help='(this build) hand fp32 views across like the reference instead of u8 frames + device-side views')
    parser.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16'],
                        help='(this build) element type of the HIP forward: bf16 (default) or fp16 -- the precision the reference '
                             'trains in on a GPU (fp16 autocast), same MFMA rate, 8x finer rounding')
    return parser.parse_args()


def main(dataset, model_dir, out_dir, model_epoch, flow_img, jitter, no_flip, host_fp32=False, dtype='bf16'):
    device = 'cuda'
    model_params = load_json(os.path.join(model_dir, 'config.json'))
    emb_dim = model_params['emb_dim']
    encoder_arch = model_params['encoder_arch']
    img_dim = model_params['img_dim']
    use_flow = model_params['use_flow']
    if use_flow:
        assert flow_img is not None, 'No flow image name specified'
    embed_time = model_params.get('embed_time', model_params.get('motion'))   # Appendix B.1
    rgb_mean_std = model_params['rgb_mean_std']
    print('Embedding dim:', emb_dim)
    print('Encoder architecture:', encoder_arch)
    print('Image dim:', img_dim)
    print('Use flow:', use_flow, '(name = {})'.format(flow_img))
    print('Embed time:', embed_time)
    print('Flip:', not no_flip)
    print('RGB mean & std:', rgb_mean_std)

    if dataset == 'tennis':
        videos, tasks = list_tennis_crops(dataset_paths.TENNIS_VIDEO_DIR, dataset_paths.TENNIS_CROP_DIR)
    else:
        videos, tasks = list_crop_dir(dataset_paths.CROPS[dataset])
    # default: decoded u8 frames cross PCIe (82 KB instead of 655 KB per frame) and the views [orig, h-flip] are built on the
    # device; --jitter needs the host fp32 views (ColorJitter on the normalised image, single_frame.py:366-379)
    raw_u8 = not host_fp32 and not jitter
    ds = FrameDataset(tasks, img_dim, rgb_mean_std, augment_jitter=jitter or 0, augment_flip=not no_flip,
                      flow_img_name=flow_img, raw_u8=raw_u8)

    model_name = 'best_epoch' if model_epoch is None else 'epoch{:04d}'.format(model_epoch)
    print('Model name:', model_name)
    encoder = RGBF_EmbeddingModel(encoder_arch, emb_dim, use_flow, device, dtype=dtype)
    encoder.load_state_dict(torch.load(os.path.join(model_dir, '{}.encoder.pt'.format(model_name)),
                                       map_location=device))
    encoder.to(device)

    loader = DataLoader(ds, batch_size=apply_batch_size(jitter, no_flip), shuffle=False,
                        num_workers=max(os.cpu_count() // 2, 1), pin_memory=True)
    # pickles are written as videos complete (the reference holds every embedding until the end, :153, :171-178)
    frames_per_video = [0] * len(videos)
    for t in tasks:
        frames_per_video[t[0]] += 1
    augmenter = None
    if raw_u8:
        from vpd_amd.augment import CropAugmenter
        augmenter = CropAugmenter(device, rgb_mean_std, img_dim, use_flow)
    embed_dataset(encoder, loader, len(videos), writer=StreamingWriter(out_dir, videos, frames_per_video),
                  augmenter=augmenter, flip=not no_flip)
    print('Done!')


if __name__ == '__main__':
    main(**vars(get_args()))
