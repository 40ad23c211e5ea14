#!/usr/bin/env python3
"""Throughput of the device input pipeline (row f1) at the bench batch: 256 u8 crops 128x128 (+flow, +mask) ->
augmented bf16 stem staging buffer (fused path) and -> fp32 NCHW batch.  HIP-event timing, resident inputs."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpd_amd import augment as A  # noqa: E402
from vpd_amd.models.rgb import RGBF_EmbeddingModel  # noqa: E402

MEAN_STD = ((0.3411329922282787, 0.46349889258964044, 0.5162481674015696),
            (0.16302619019820488, 0.17092395707914718, 0.19266662199338647))
N, HW = 256, 128
rs = np.random.RandomState(0)
rgb = torch.from_numpy(rs.randint(0, 256, (N, HW, HW, 3)).astype(np.uint8)).cuda()
flow = torch.from_numpy(rs.randint(0, 256, (N, HW, HW, 2)).astype(np.uint8)).cuda()
mask = torch.from_numpy((rs.rand(N, HW, HW) > 0.5).astype(np.uint8) * 255).cuda()
params = A.sample_params(N, HW, HW, generator=torch.Generator().manual_seed(0))
enc = RGBF_EmbeddingModel("resnet34", 128, True, torch.device("cuda:0"))
eng = enc.engine
aug = A.CropAugmenter("cuda:0", MEAN_STD, HW, True)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


out = torch.empty((N, 5, HW, HW), device="cuda")
t_stage = timeit(lambda: aug.stage(eng, rgb, flow, mask, params, train=True))
t_f32 = timeit(lambda: aug(rgb, flow, mask, params, out=out))
alg_stage = N * HW * HW * (6 + 16)            # 5 B u8 + 1 B mask read, 8 x bf16 written per pixel
alg_f32 = N * HW * HW * (6 + 20)
print(json.dumps({"crops": N, "stage_us": t_stage, "stage_crops_per_s": N / t_stage * 1e6,
                  "stage_GBps_algorithmic": alg_stage / t_stage / 1e3,
                  "fp32_batch_us": t_f32, "fp32_GBps_algorithmic": alg_f32 / t_f32 / 1e3,
                  "note": "includes the host-side parameter upload (16 KB) and argument checks of every call"}))
