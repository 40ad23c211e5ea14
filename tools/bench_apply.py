#!/usr/bin/env python3
"""Inference twin of the hot path (BASELINE.json configs[4], SURVEY.md 8d "C5"): apply_vpd_model.py's loop on the
HIP engine -- batches of 500 frames x 2 views (orig + h-flip) = 1000 crops of 5x128x128, eval-mode ResNet-34
forward as ONE hipGraph launch per batch, embeddings to the host once per batch, per-video tuple lists.

Prints one JSON line with three rates (crops/s):
  forward_resident    graph launches only, input batch resident in HBM (kernel-side rate)
  loop_resident       vpd_amd.apply.embed_dataset on device-resident batches (adds D2H of embeddings + list assembly)
  loop_host_fp32      the same loop fed from pinned host fp32 batches (adds the 327,680 B/crop H2D copy: PCIe-bound)

  python tools/bench_apply.py [--batches 20]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ARCH, C_IN, EMB_DIM, HW, FRAMES, K = "resnet34", 5, 128, 128, 500, 2
FWD_FLOP_PER_CROP = 2443837440


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    from vpd_amd.apply import embed_dataset
    from vpd_amd.models.rgb import RGBF_EmbeddingModel

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    enc = RGBF_EmbeddingModel(ARCH, EMB_DIM, True, dev)
    enc.reset_parameters(seed=0)
    enc.eval()
    eng = enc.engine
    n = FRAMES * K
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((FRAMES, K, C_IN, HW, HW), generator=g, device=dev)

    # ---- 1. graph launches only ----
    xin = x.reshape(n, C_IN, HW, HW).contiguous()
    out = torch.empty((n, EMB_DIM), dtype=torch.float32, device=dev)
    pl = eng.capture_eval_graph(xin, out)
    for _ in range(args.warmup):
        eng.launch_eval_graph(pl, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.batches):
        eng.launch_eval_graph(pl, n)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t0

    # ---- 2./3. the apply loop (Diving48-shaped: ~157 frames per video) ----
    frames_per_video = 157
    n_videos = (args.batches * FRAMES + frames_per_video - 1) // frames_per_video

    def loader(img):
        f = 0
        for _ in range(args.batches):
            idx = torch.arange(f, f + FRAMES)
            yield {"video": (idx // frames_per_video), "frame": (idx % frames_per_video), "img": img}
            f += FRAMES

    def timed(img):
        embed_dataset(enc, ({"video": torch.zeros(FRAMES, dtype=torch.long), "frame": torch.arange(FRAMES), "img": img}
                            for _ in range(args.warmup)), 1)
        torch.cuda.synchronize()
        t = time.perf_counter()
        embs = embed_dataset(enc, loader(img), n_videos)
        torch.cuda.synchronize()
        return time.perf_counter() - t, embs

    dt2, embs = timed(x)
    xh = x.cpu().pin_memory()
    dt3, _ = timed(xh)
    crops = args.batches * n
    assert sum(len(v) for v in embs) == args.batches * FRAMES
    res = {"metric": "frame-crops/sec (VPD student apply, eval forward)", "unit": "crops/s",
           "workload": "configs[4]-shaped: %d batches of %d frames x %d views, ResNet-34 5x128x128, D=%d, bf16, hipGraph"
                       % (args.batches, FRAMES, K, EMB_DIM),
           "forward_resident": crops / dt1, "loop_resident": crops / dt2, "loop_host_fp32": crops / dt3,
           "ms_per_batch_forward": 1e3 * dt1 / args.batches,
           "forward_tflops": crops / dt1 * FWD_FLOP_PER_CROP / 1e12,
           "h2d_GBps_in_loop": crops * C_IN * HW * HW * 4 / dt3 / 1e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
