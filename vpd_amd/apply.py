"""Inference loop of reference apply_vpd_model.py:146-178 on the HIP engine:
batches of BATCH_SIZE frames x k views -> embed -> per-video sorted list of
(frame:int, f32[k,D] or f32[D], {}) -> <video>.emb.pkl.  The eval forward is
hipGraph-captured per batch shape; embeddings leave the GPU once per batch."""
import os

import torch

from .io import store_pickle

BATCH_SIZE = 500          # apply_vpd_model.py:15


def apply_batch_size(jitter, no_flip):
    bs = BATCH_SIZE
    if jitter is not None:
        bs = bs // (jitter + 1)
    if no_flip:
        bs *= 2
    return bs


def embed_dataset(encoder, loader, n_videos, progress_cb=None, use_graph=True):
    """loader yields {'video': int[n], 'frame': int[n], 'img': f32[n,k,C,H,W]} -> list per video."""
    eng = encoder.engine
    encoder.eval()
    all_embs = [list() for _ in range(n_videos)]
    graphs = {}
    for batch in loader:
        video_ids = batch['video'].tolist() if hasattr(batch['video'], 'tolist') else list(batch['video'])
        frame_nums = batch['frame'].tolist() if hasattr(batch['frame'], 'tolist') else list(batch['frame'])
        n_batch, k, c, h, w = batch['img'].shape
        x = batch['img'].reshape(-1, c, h, w)
        if encoder.use_flow:
            assert c == 5, 'Wrong number of channels for RGB + flow'
        else:
            assert c == 3, 'Wrong number of channels for RGB'
        n = x.shape[0]
        if use_graph:
            if n not in graphs:          # one captured graph per batch shape (full batches + the tail)
                xin = torch.empty((n, c, h, w), dtype=torch.float32, device=eng.device)
                out = torch.empty((n, encoder.emb_dim), dtype=torch.float32, device=eng.device)
                graphs[n] = (eng.capture_eval_graph(xin, out), xin, out)
            pl, xin, out = graphs[n]
            xin.copy_(x, non_blocking=True)
            eng.launch_eval_graph(pl, n)
            embs = out.cpu().numpy()
        else:
            embs = encoder.embed(x)
        embs = embs.reshape((n_batch, k, -1))
        for i in range(n_batch):
            all_embs[video_ids[i]].append((frame_nums[i], embs[i, :, :].copy() if k > 1 else embs[i, 0, :].copy(), {}))
        if progress_cb is not None:
            progress_cb(n_batch)
    return all_embs


def write_embeddings(out_dir, videos, all_embs):
    for video_name, embs in zip(videos, all_embs):
        if len(embs) > 0:
            embs.sort(key=lambda t: t[0])
            os.makedirs(out_dir, exist_ok=True)
            store_pickle(os.path.join(out_dir, '{}.emb.pkl'.format(video_name)), embs)
