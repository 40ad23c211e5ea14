"""Inference loop of reference apply_vpd_model.py:146-178 on the HIP engine:
batches of BATCH_SIZE frames x k views -> embed -> per-video sorted list of
(frame:int, f32[k,D] or f32[D], {}) -> <video>.emb.pkl.  The eval forward is
hipGraph-captured per batch shape; embeddings leave the GPU once per batch."""
import gc
import os
import pickle

import numpy as np
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


class StreamingWriter:
    """Row f4: write <video>.emb.pkl as soon as the video's last frame is embedded instead of holding every
    embedding of the data set in RAM until the end (reference apply_vpd_model.py:153, :171-178 holds 1 M of them).
    Same files, same content: a list sorted by frame of (frame:int, f32[k,D] or f32[D], {})."""

    def __init__(self, out_dir, videos, frames_per_video):
        self.out_dir, self.videos = out_dir, list(videos)
        self.remaining = list(frames_per_video)
        self.pending = [list() for _ in self.videos]
        self.written = []
        self._flushed = set()
        if out_dir is not None:
            os.makedirs(out_dir, exist_ok=True)

    def add(self, video_id, item):
        self.pending[video_id].append(item)
        self.remaining[video_id] -= 1
        if self.remaining[video_id] == 0:
            self.flush(video_id)

    def add_many(self, video_id, items):
        """A run of one video's frames out of one batch (embed_dataset hands over runs, not frames)."""
        self.pending[video_id].extend(items)
        self.remaining[video_id] -= len(items)
        if self.remaining[video_id] <= 0:
            self.flush(video_id)

    def flush(self, video_id):
        embs = self.pending[video_id]
        if embs and self.out_dir is not None:
            path = os.path.join(self.out_dir, '{}.emb.pkl'.format(self.videos[video_id]))
            if self.videos[video_id] in self._flushed:
                # frames of a video whose pickle is already on disk (its frame count was under-estimated): the file would be
                # overwritten with the late frames only -- merge with what was written instead of losing it silently
                with open(path, 'rb') as fp:
                    embs = pickle.load(fp) + embs
            embs.sort(key=lambda t: t[0])
            store_pickle(path, embs)
            if self.videos[video_id] not in self._flushed:
                self.written.append(self.videos[video_id])
            self._flushed.add(self.videos[video_id])
        self.pending[video_id] = []

    def close(self):
        for vid in range(len(self.videos)):      # videos whose frame count was over-estimated
            if self.pending[vid]:
                self.flush(vid)


def embed_dataset(encoder, loader, n_videos, progress_cb=None, use_graph=True, writer=None, augmenter=None, flip=True):
    """loader yields {'video': int[n], 'frame': int[n], 'img': f32[n,k,C,H,W]} -> list per video, or, with
    `writer` (a StreamingWriter), pickles written as videos complete (returns None).

    Raw batches {'video', 'frame', 'rgb_u8': u8[n,H,W,3], 'flow_u8': u8[n,H,W,2]} (FrameDataset(raw_u8=True)) need
    `augmenter` (a vpd_amd.augment.CropAugmenter with the model's mean / std): 82 KB per frame cross PCIe instead of the
    655 KB of two fp32 views, and the views [orig, h-flip] (`flip`) are built on the device in the stem's staging buffer
    (same values as FrameDataset's fp32 views, vpd_dataset/single_frame.py:377-400).

    The embeddings of batch i travel to a pinned host buffer asynchronously; the host turns batch i-1 into tuples
    (and pickles) while the GPU runs batch i."""
    eng = encoder.engine
    encoder.eval()
    if os.environ.get("VPD_APPLY_GRAPH", "1") == "0":
        use_graph = False
    all_embs = None if writer is not None else [list() for _ in range(n_videos)]
    # captured graphs (and the pinned staging buffers) live on the engine: a second call with the same batch shape replays
    # them instead of paying capture + instantiation again (tens of milliseconds: as much as several 1,000-crop batches)
    graphs = eng.__dict__.setdefault("_apply_graphs", {})
    host = eng.__dict__.setdefault("_apply_host", {})            # (n, slot) -> pinned host buffer

    def drain(job):
        ev, hbuf, video_ids, frame_nums, n_batch, k = job
        ev.synchronize()
        # ONE copy of the batch out of the pinned buffer; a frame's embedding is a view of it (pickled, a view stores just its
        # own bytes).  The (frame, emb, {}) tuples of the reference's format (apply_vpd_model.py:166-169) come out of one zip
        # and travel per run of equal video ids, not frame by frame: per-frame copy / tuple / add() calls were 0.35 ms of host
        # time per 500-frame batch.
        block = hbuf.numpy().reshape((n_batch, k, -1)).copy()
        items = list(zip(frame_nums, block if k > 1 else block[:, 0, :], [{} for _ in range(n_batch)]))
        vid = np.asarray(video_ids)
        cuts = [0] + (np.flatnonzero(vid[1:] != vid[:-1]) + 1).tolist() + [n_batch] if n_batch else [0]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if writer is not None:
                writer.add_many(video_ids[a], items[a:b])
            else:
                all_embs[video_ids[a]].extend(items[a:b])
        if progress_cb is not None:
            progress_cb(n_batch)

    # The loop creates three container objects per frame (tuple, view, dict) that cannot form cycles, while a FULL pass of the
    # cyclic collector over a torch-sized heap takes 75-90 ms (measured: one such pass inside a 12-batch run turns 270 k
    # crops/s into 95 k) with the GPU idle behind it.  gc.freeze() parks everything allocated so far in the permanent
    # generation: the collector stays ON for the whole job (hours for a large data set: cycles made by the loader, its
    # workers or the writer are still collected), its passes only walk what the job itself has allocated since.
    # (a freeze the CALLER made is the caller's to undo: only our own is lifted)
    ours = gc.get_freeze_count() == 0
    if ours:
        gc.freeze()
    try:
        return _embed_loop(encoder, eng, loader, graphs, host, drain, writer, all_embs, augmenter, flip, use_graph)
    finally:
        if ours:
            gc.unfreeze()


def _embed_loop(encoder, eng, loader, graphs, host, drain, writer, all_embs, augmenter, flip, use_graph):
    inflight = None      # (event, host buffer, video_ids, frame_nums, n_batch, k)
    slot = 0
    for batch in loader:
        video_ids = batch['video'].tolist() if hasattr(batch['video'], 'tolist') else list(batch['video'])
        frame_nums = batch['frame'].tolist() if hasattr(batch['frame'], 'tolist') else list(batch['frame'])
        if 'rgb_u8' in batch:
            if augmenter is None:
                raise RuntimeError("raw u8 batches need embed_dataset(..., augmenter=CropAugmenter(...))")
            assert (batch.get('flow_u8') is not None) == bool(encoder.use_flow), 'Wrong number of channels'
            n_batch, k = batch['rgb_u8'].shape[0], (2 if flip else 1)
            # H2D on a copy stream into one of two device slots: batch i + 1 crosses PCIe while batch i's forward runs
            cur = torch.cuda.current_stream(eng.device)
            cp = eng.__dict__.setdefault("_apply_copy_stream", None) or torch.cuda.Stream(device=eng.device)
            eng._apply_copy_stream = cp
            ubuf = eng.__dict__.setdefault("_apply_u8", {})
            uslot = eng.__dict__.get("_apply_u8_slot", 0) ^ 1
            eng._apply_u8_slot = uslot
            ukey = (tuple(batch['rgb_u8'].shape), uslot)
            if ukey not in ubuf:
                ubuf[ukey] = [torch.empty(tuple(batch['rgb_u8'].shape), dtype=torch.uint8, device=eng.device),
                              torch.empty(tuple(batch['rgb_u8'].shape[:3]) + (2,), dtype=torch.uint8, device=eng.device)
                              if encoder.use_flow else None, None]
            rgb, flow, freed = ubuf[ukey]
            with torch.cuda.stream(cp):
                if freed is not None:
                    cp.wait_event(freed)             # the staging launch that last read this slot is done
                rgb.copy_(batch['rgb_u8'], non_blocking=True)
                if flow is not None:
                    flow.copy_(batch['flow_u8'], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(cp)
            cur.wait_event(landed)
            n, hw = augmenter.stage_views(eng, rgb, flow, flip)
            ubuf[ukey][2] = torch.cuda.Event()
            ubuf[ukey][2].record(cur)
            key = ('staged', n, hw)
            ent = graphs.get(key) if use_graph else None
            if ent is not None and not (ent[0].handle and n in ent[0].graph_sizes and eng._plans.get((hw, hw, False, False)) is ent[0]):
                ent = None
            if use_graph:
                if ent is None:
                    out = torch.empty((n, encoder.emb_dim), dtype=torch.float32, device=eng.device)
                    ent = graphs[key] = (eng.capture_eval_graph_staged(n, hw, out), None, out)
                    augmenter.stage_views(eng, rgb, flow, flip)      # (a larger plan may have been built: stage again)
                    ubuf[ukey][2].record(cur)
                pl, _, out = ent
                eng.launch_eval_graph(pl, n)
            else:
                out = eng.forward_eval(None, staged=(n, hw))
            slot ^= 1
            if (n, slot) not in host:
                host[(n, slot)] = torch.empty((n, encoder.emb_dim), dtype=torch.float32).pin_memory()
            hbuf = host[(n, slot)]
            hbuf.copy_(out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(eng.device))
            if inflight is not None:
                drain(inflight)
            inflight = (ev, hbuf, video_ids, frame_nums, n_batch, k)
            continue
        n_batch, k, c, h, w = batch['img'].shape
        x = batch['img'].reshape(-1, c, h, w)
        if encoder.use_flow:
            assert c == 5, 'Wrong number of channels for RGB + flow'
        else:
            assert c == 3, 'Wrong number of channels for RGB'
        n = x.shape[0]
        if use_graph:
            key = (n, c, h, w)
            ent = graphs.get(key)
            if ent is not None and not (ent[0].handle and n in ent[0].graph_sizes and eng._plans.get((h, w, False, False)) is ent[0]):
                ent = None               # the plan was rebuilt (a larger batch came by): its graphs went with it
            if ent is None:              # one captured graph per batch shape (full batches + the tail)
                xin = torch.empty((n, c, h, w), dtype=torch.float32, device=eng.device)
                out = torch.empty((n, encoder.emb_dim), dtype=torch.float32, device=eng.device)
                ent = graphs[key] = (eng.capture_eval_graph(xin, out), xin, out)
            pl, xin, out = ent
            xin.copy_(x, non_blocking=True)
            eng.launch_eval_graph(pl, n)
        else:
            out = eng.forward_eval(x.to(eng.device, dtype=torch.float32).contiguous())
        slot ^= 1
        if (n, slot) not in host:
            host[(n, slot)] = torch.empty((n, encoder.emb_dim), dtype=torch.float32).pin_memory()
        hbuf = host[(n, slot)]
        hbuf.copy_(out, non_blocking=True)          # stream order: after the forward, before the next one rewrites `out`
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(eng.device))
        if inflight is not None:
            drain(inflight)                          # host work of the previous batch overlaps this batch's forward
        inflight = (ev, hbuf, video_ids, frame_nums, n_batch, k)
    if inflight is not None:
        drain(inflight)
    if writer is not None:
        writer.close()
    return all_embs


def write_embeddings(out_dir, videos, all_embs):
    for video_name, embs in zip(videos, all_embs):
        if len(embs) > 0:
            embs.sort(key=lambda t: t[0])
            os.makedirs(out_dir, exist_ok=True)
            store_pickle(os.path.join(out_dir, '{}.emb.pkl'.format(video_name)), embs)
