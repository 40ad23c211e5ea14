"""Reader half of row f4: the downstream consumer of `<video>.emb.pkl` (reference action_dataset/load.py:16-64).

`group_by_frame` turns one video's sparse list of (frame, emb, meta) into a dense [num_frames, ...] array: embeddings
that share a frame number (tennis: several crops of one frame) are AVERAGED (load.py:24-32), frames without an
embedding between two embedded frames are filled by the reference's linear blend (load.py:34-42; note its weights:
the frame i steps after `prev` gets a = i / gap on `prev` and 1 - a on the next embedded frame -- kept as it is), frames
before the first embedded one stay zero, and the second return value marks the frames that had an embedding.
`load_embs` applies it to every pickle of a directory, optionally L2-normalising rows (load.py:46-64).

Written over whole arrays (one scatter-add, one division, one blend per gap) instead of per-frame Python statements;
results are bit-identical to the reference's loops (same float64 operations in the same order per element), which
tests/test_host_cpu.py checks against arrays the reference itself produced (tests/golden/format_case.npz,
tests/golden/reader_case.npz)."""
import os

import numpy as np

from .io import load_pickle


def group_by_frame(embs):
    """embs: list of (frame:int, emb f32[D] or f32[K, D], meta) -> (dense f64[num_frames, D] or [num_frames, K, D], mask)."""
    frame_of = np.fromiter((t[0] for t in embs), dtype=np.int64, count=len(embs))
    first = np.asarray(embs[0][1])
    # (load.py:18-22: a [K, D] embedding keeps its shape, anything else contributes its last axis)
    cell = first.shape if first.ndim == 2 else (first.shape[-1],)
    num_frames = int(frame_of.max()) + 1
    dense = np.zeros((num_frames,) + cell)
    vals = np.stack([np.asarray(t[1], dtype=np.float64).reshape(cell) for t in embs])
    np.add.at(dense, frame_of, vals)                 # unbuffered and in list order: the sums of the reference's loop
    counts = np.bincount(frame_of, minlength=num_frames).astype(np.float64)
    seen = counts > 0
    dense[seen] /= counts[seen].reshape((-1,) + (1,) * len(cell))
    frames = np.flatnonzero(seen)
    for prev, nxt in zip(frames[:-1], frames[1:]):
        gap = int(nxt - prev)
        if gap > 1:
            a = (np.arange(1, gap) / gap).reshape((-1,) + (1,) * len(cell))
            dense[prev + 1:nxt] = a * dense[prev] + (1. - a) * dense[nxt]
    return dense, seen


def normalize_rows(x):
    """L2-normalise the last axis of [T, D] (or [T, K, D]) rows; rows shorter than 1e-12 are left alone (load.py:46-49)."""
    d = np.linalg.norm(x, axis=1 if x.ndim == 2 else 2, keepdims=True)
    d[d < 1e-12] = 1
    return x / d


def load_embs(emb_dir, norm, emb_ext='.emb.pkl'):
    """{video name: (dense, mask)} for every `<video><emb_ext>` of emb_dir (load.py:52-64)."""
    print('Loading embs:', emb_dir)
    emb_dict = {}
    for emb_file in os.listdir(emb_dir):
        if emb_file.endswith(emb_ext):
            dense, mask = group_by_frame(load_pickle(os.path.join(emb_dir, emb_file)))
            emb_dict[emb_file[:-len(emb_ext)]] = (normalize_rows(dense) if norm else dense, mask)
    print('  shape:', list(emb_dict.values())[0][0].shape[1:])
    return emb_dict
