#!/usr/bin/env python3
"""Drop-in for reference train_vpd_model.py (same flags, files and bookkeeping; the
student's arithmetic runs on libvpdhip).  Extra, optional flags: --synthetic N (no
dataset files needed) and torchrun for multi-GPU data parallelism."""
import argparse
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from vpd_amd import paths as dataset_paths
from vpd_amd.data import RGB_MEAN_STD, SyntheticCrops, TeacherEmbDataset, load_tennis_default
from vpd_amd.io import store_json
from vpd_amd.models.rgb import RGBF_EmbeddingModel
from vpd_amd.trainer import ModelTrainer

DATASETS = ['tennis', 'fs', 'fx', 'penn', 'diving48']


def get_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('dataset', type=str, choices=DATASETS)
    parser.add_argument('--save_dir', type=str, required=True)
    parser.add_argument('--checkpoint_frequency', type=int)
    parser.add_argument('--num_epochs', type=int, default=1000)
    parser.add_argument('--batch_size', type=int, default=100)
    parser.add_argument('--learning_rate', type=float, default=0.0005)
    parser.add_argument('--img_dim', type=int, default=128)
    parser.add_argument('--flow_img', type=str)
    parser.add_argument('--motion', action='store_true')
    parser.add_argument('--encoder_arch', type=str, default='resnet34')
    parser.add_argument('--model_select_window', type=int, default=5)
    parser.add_argument('--pretrained', action='store_true')
    parser.add_argument('--no_test_video', action='store_true')
    parser.add_argument('--min_pose_score', type=float)
    dataset_group = parser.add_mutually_exclusive_group()
    dataset_group.add_argument('--emb_dir', type=str)
    dataset_group.add_argument('--penn_dir', type=str)
    # extensions (not in the reference)
    parser.add_argument('--synthetic', type=int, help='train on N seeded synthetic crops per epoch')
    parser.add_argument('--synthetic_emb_dim', type=int, default=128)
    parser.add_argument('--gpu_augment', action='store_true',
                        help='(default behaviour, kept for compatibility) loaders hand over decoded u8 crops; ColorJitter / '
                             'mask noise / RandomResizedCrop / normalisation run on the GPU (vpd_amd/augment.py)')
    parser.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16'],
                        help='(this build) element type of the HIP path: bf16 (default, no loss scaling) or fp16 -- the reference\'s own '
                             'GPU precision (fp16 autocast + GradScaler), here with a static loss scale (vpd_amd.models.util.LossScaler)')
    parser.add_argument('--no_augment', action='store_true',
                        help='escape hatch: fp32 batches from the CPU loaders with h-flips only -- NOT the reference '
                             'recipe, whose datasets always augment (vpd_dataset/common.py:85-92)')
    return parser.parse_args()


def get_moving_avg_loss(losses, n, key):
    return np.mean([l[key] for l in losses[-n:]])


def load_dataset(dataset, dataset_kwargs, emb_dir, penn_dir, no_test_video):
    if dataset == 'penn':
        raise NotImplementedError('PennDataset reads a hard-coded private path in the reference (out of scope)')
    if no_test_video:          # hold the downstream test videos out of distillation (train_vpd_model.py:125-156)
        from vpd_amd.splits import get_test_prefixes
        dataset_kwargs['exclude_prefixes'] = get_test_prefixes(dataset)
    if emb_dir is None:
        emb_dir = os.path.join(dataset_paths.ROOT[dataset], 'embs')
    if dataset == 'tennis':          # per-player crop directories, split over clips (train_vpd_model.py:121-128)
        return load_tennis_default(emb_dir, dataset_paths.CROPS[dataset], **dataset_kwargs)
    return TeacherEmbDataset.load_default(emb_dir, dataset_paths.CROPS[dataset], **dataset_kwargs)


def main(dataset, num_epochs, batch_size, learning_rate, img_dim, flow_img, motion, encoder_arch, save_dir,
         model_select_window, checkpoint_frequency, pretrained, emb_dir, penn_dir, no_test_video, min_pose_score,
         synthetic=None, synthetic_emb_dim=128, gpu_augment=False, no_augment=False, dtype='bf16'):
    device = 'cuda'
    # The reference builds train AND val datasets with augment=True (vpd_dataset/single_frame.py:267-272,
    # common.py:85-92): ColorJitter, mask noise, RandomResizedCrop and flips are part of the recipe, so they are the
    # default here too and run on the device; --no_augment is the only way to switch them off.
    gpu_augment = not no_augment
    rank, world = 0, 1
    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        torch.distributed.init_process_group(os.environ.get('VPD_DIST_BACKEND', 'nccl'))      # 'nccl' = RCCL
    # save_dir must not exist (train_vpd_model.py:221).  Rank 0 creates it FIRST and tells the others, so that every
    # rank leaves together instead of waiting in a collective for a rank that has raised
    made = [None, int.from_bytes(os.urandom(4), 'little')]       # [makedirs error or None, split seed]
    if rank == 0:
        try:
            os.makedirs(save_dir)
        except OSError as e:
            made[0] = repr(e)
    split_seed = None
    if world > 1:
        # one train/val split for all ranks (the reference's train_test_split is unseeded, vpd_dataset/
        # single_frame.py:263: a split per rank would train the shared weights on other ranks' validation frames)
        torch.distributed.broadcast_object_list(made, src=0)
        split_seed = made[1]
    if made[0] is not None:
        if world > 1:
            torch.distributed.destroy_process_group()
        raise FileExistsError('cannot create save_dir {!r}: {}'.format(save_dir, made[0]))
    rgb_mean_std = RGB_MEAN_STD['resnet' if pretrained else dataset]

    if synthetic is not None:
        use_flow = flow_img is not None
        emb_dim = synthetic_emb_dim
        mk = lambda n, seed: SyntheticCrops(n, 5 if use_flow else 3, img_dim, emb_dim, motion, rgb_mean_std, seed,
                                            raw_u8=gpu_augment)
        train_dataset, val_dataset = mk(synthetic, 1 + rank), mk(max(synthetic // 5, 1), 1001 + rank)
    else:
        dataset_kwargs = {'img_dim': img_dim, 'flow_img_name': flow_img, 'embed_time': motion,
                          'rgb_mean_std': rgb_mean_std, 'target_len': 20000 // world, 'split_seed': split_seed}
        if min_pose_score is not None:
            dataset_kwargs['min_pose_score'] = min_pose_score
        train_dataset, val_dataset, emb_dim = load_dataset(dataset, dataset_kwargs, emb_dir, penn_dir, no_test_video)
        train_dataset.raw_u8 = val_dataset.raw_u8 = gpu_augment
        train_dataset.augment = val_dataset.augment = True      # flips stay on with --no_augment (the CPU float path)

    if rank == 0:
        print('Device:', device)
        print('Num epochs:', num_epochs)
        print('Batch size:', batch_size)
        print('Image dim:', img_dim)
        print('Use flow:', flow_img is not None)
        print('Embed time:', motion)
        print('Encoder arch:', encoder_arch)
        print('Dataset:')
        print('', 'Train images:', len(train_dataset))
        print('', 'Val images:', len(val_dataset))
        print('', 'Embedding dim:', emb_dim)
        print('', 'Min pose score:', min_pose_score)

    num_load_workers = min(os.cpu_count(), 8)
    train_loader = DataLoader(train_dataset, batch_size, shuffle=True, num_workers=num_load_workers,
                              persistent_workers=False, pin_memory=True)
    val_loader = DataLoader(val_dataset, batch_size, num_workers=num_load_workers, persistent_workers=False,
                            pin_memory=True)

    encoder = RGBF_EmbeddingModel(encoder_arch, emb_dim, flow_img is not None, device, pretrained=pretrained, dtype=dtype)
    augmenter = None
    if gpu_augment:
        from vpd_amd.augment import CropAugmenter
        augmenter = CropAugmenter(encoder.device, rgb_mean_std, img_dim, flow_img is not None)
    trainer = ModelTrainer(encoder, motion, augmenter=augmenter, augment=True)
    if world > 1:      # same initial weights on every rank (after the trainer: the motion head is initialised there)
        torch.distributed.broadcast(encoder.engine.params, 0)
        torch.distributed.broadcast(encoder.engine.bn_running, 0)
        encoder.engine.mark_weights_changed()
    optimizer, scaler = trainer.get_optimizer(learning_rate)

    if rank == 0:
        store_json(os.path.join(save_dir, 'config.json'), {
            'num_epochs': num_epochs, 'batch_size': batch_size, 'learning_rate': learning_rate, 'img_dim': img_dim,
            'use_flow': flow_img is not None, 'motion': motion,
            'embed_time': motion,      # apply_vpd_model.py:102 reads this key; the reference never writes it
            'emb_dim': emb_dim, 'encoder_arch': encoder_arch, 'rgb_mean_std': rgb_mean_std,
            'dtype': dtype,            # not in the reference: element type of the HIP path (bf16 | fp16 + static loss scale)
            # not in the reference: which input pipeline produced the loss curves
            'augment': 'device: ColorJitter + mask noise + RandomResizedCrop + flip (reference recipe)' if gpu_augment
                       else 'cpu: h-flip only (--no_augment)'})

    loss_file = os.path.join(save_dir, 'loss.json')
    losses = []
    best_val_loss = float('inf')
    for epoch in range(1, num_epochs + 1):
        train_loss = trainer.epoch(train_loader, optimizer=optimizer, scaler=scaler)
        if world > 1 and os.environ.get('VPD_DDP_AVG_BN', '0') == '1':      # per-rank BatchNorm statistics -> their mean over ranks
            from vpd_amd.ddp import average_running_stats
            average_running_stats(encoder.engine.bn_running)
        val_loss = trainer.epoch(val_loader)
        losses.append({'epoch': epoch, 'train': train_loss, 'val': val_loss,
                       'dataset_train': [(dataset, train_loss)], 'dataset_val': [(dataset, val_loss)]})
        moving_avg_val_loss = get_moving_avg_loss(losses, model_select_window, 'val')
        if rank == 0:
            print('Epoch {} - train loss: {:0.4f} [avg: {:0.4f}] val loss: {:0.4f} [avg: {:0.4f}]'.format(
                epoch, train_loss, get_moving_avg_loss(losses, model_select_window, 'train'), val_loss,
                moving_avg_val_loss))
            store_json(loss_file, losses)
            if moving_avg_val_loss < best_val_loss:
                print('New best epoch!')
                trainer.save_model(save_dir, 'best_epoch')
            if checkpoint_frequency is not None and epoch % checkpoint_frequency == 0:
                print('Saving checkpoint: {}'.format(epoch))
                trainer.save_model(save_dir, 'epoch{:04d}'.format(epoch))
        best_val_loss = min(moving_avg_val_loss, best_val_loss)
    if rank == 0:
        print('Saving last epoch: {}'.format(epoch))
        trainer.save_model(save_dir, 'epoch{:04d}'.format(epoch))
        print('Done!')
    if world > 1:
        # replicas must hold identical weights (same initial broadcast, same summed gradients, same AdamW): verify
        chk = torch.stack([encoder.engine.params.double().sum(), encoder.engine.params.double().square().sum()])
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        # (a one-shot all-reduce may sum in a rank-dependent order: ulp-level drift is tolerated, a missed
        # broadcast or a skipped bucket is not)
        if bool(((hi - lo).abs() > 1e-4 * hi.abs().clamp_min(1e-30)).any()):
            raise RuntimeError('data-parallel replicas diverged: parameter checksums {} .. {}'.format(
                lo.tolist(), hi.tolist()))
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main(**vars(get_args()))
