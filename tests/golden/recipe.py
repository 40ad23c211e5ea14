"""Seed recipes of the committed golden fixtures (numpy only; imports nothing of this repo).

The fixtures under tests/golden/ hold what the REFERENCE produced (oracle/gen_golden.py ran its RGBF_EmbeddingModel /
ModelTrainer on these inputs); the inputs themselves -- weights and crops -- are too big to commit and are regenerated bit for bit
from the seeds in each fixture's `meta`.  This file restates the two generators so that `bench.py`'s parity block can
rebuild a fixture's inputs without touching `oracle/`; tests/test_oracle_golden.py checks that they produce the
same bits as oracle.vpd_oracle.procedural_state_dict / synthetic_crops.
"""
import math
from collections import OrderedDict

import numpy as np

DIVING48_MEAN_STD = ((0.3411329922282787, 0.46349889258964044, 0.5162481674015696),
                     (0.16302619019820488, 0.17092395707914718, 0.19266662199338647))


def _kind(key, shape):
    if key.endswith("num_batches_tracked"):
        return "bn_nbt"
    if len(shape) == 4:
        return "conv"
    if len(shape) == 2:
        return "fc_w"
    if key.endswith("running_mean"):
        return "bn_rm"
    if key.endswith("running_var"):
        return "bn_rv"
    if ".fc." in key or key.startswith("layers."):
        return "fc_b"
    return "bn_w" if key.endswith("weight") else "bn_b"


def procedural_weights(shapes, seed):
    """shapes: ordered {state_dict key: shape} in the reference's module order -> {key: float32 array (int64 scalar for
    num_batches_tracked)}; every tensor in key order from ONE RandomState(seed)."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        kind = _kind(key, shape)
        if kind == "conv":
            co, _, kh, kw = shape
            a = rs.standard_normal(shape) * math.sqrt(2.0 / (co * kh * kw))
        elif kind == "bn_w":
            a = rs.uniform(0.5, 1.5, shape)
        elif kind in ("bn_b", "bn_rm"):
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            a = rs.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            out[key] = np.zeros((), dtype=np.int64)
            continue
        elif kind == "fc_w":
            bound = 1.0 / math.sqrt(shape[1])
            a = rs.uniform(-bound, bound, shape)
        else:
            a = rs.uniform(-0.04, 0.04, shape)
        out[key] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def synthetic_crops(n, c_in, hw, seed, mean_std=None):
    """fp32 NCHW crops in the reference's value ranges (vpd_dataset/common.py:52-69): normalised RGB of uniform u8 draws, flow
    planes clip(round(124 + 12 N(0,1))) / 255 - 0.5."""
    rs = np.random.RandomState(seed)
    rgb = rs.randint(0, 256, size=(n, 3, hw, hw)).astype(np.float32) / 255.0
    mean_std = mean_std or DIVING48_MEAN_STD
    mean = np.asarray(mean_std[0], np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(mean_std[1], np.float32).reshape(1, 3, 1, 1)
    x = (rgb - mean) / std
    if c_in > 3:
        fl = np.clip(np.round(124 + 12 * rs.standard_normal((n, c_in - 3, hw, hw))), 0, 255)
        x = np.concatenate([x, fl.astype(np.float32) / 255.0 - 0.5], axis=1)
    return np.ascontiguousarray(x, dtype=np.float32)
