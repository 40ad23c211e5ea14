#!/usr/bin/env python3
"""Inference twin of the hot path (BASELINE.json configs[4], SURVEY.md 8d "C5"): apply_vpd_model.py's loop on the
HIP engine -- batches of 500 frames x 2 views (orig + h-flip) = 1000 crops of 5x128x128, eval-mode ResNet-34
forward as ONE hipGraph launch per batch, embeddings to the host once per batch, per-video tuple lists.

Prints one JSON line with rates in crops/s:
  forward_resident    graph launches only, input batch resident in HBM (kernel-side rate)
  loop_resident       vpd_amd.apply.embed_dataset on device-resident batches (adds D2H of embeddings + list assembly)
  loop_host_fp32      the same loop fed from pinned host fp32 batches (adds the 327,680 B/crop H2D copy: PCIe-bound)
  loop_host_u8        the loop fed from pinned host u8 FRAMES (81,920 B per frame = 2 views; normalise / flow decode / h-flip
                      on the device, straight into the stem's staging buffer: FrameDataset(raw_u8=True))
With --crops N (configs[4] is 1,000,000): the whole job -- N / 2 frames of Diving48-shaped videos (18,404 videos per
1 M crops, 25 ... 823 frames each, mean 157: SURVEY 8d), u8 frames from pinned host memory, StreamingWriter flushing a real
<video>.emb.pkl per video into --out_dir (apply_vpd_model.py:146-178) -- as `full_run`.

  python tools/bench_apply.py [--batches 20] [--crops 1000000 --out_dir /tmp/vpd_apply_out]
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
    ap.add_argument("--crops", type=int, default=0, help="full job of this many crops (2 views per frame) with real pickles")
    ap.add_argument("--out_dir", default="/tmp/vpd_apply_out")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="element type of the HIP forward (apply_vpd_model.py --dtype)")
    args = ap.parse_args()
    from vpd_amd.apply import StreamingWriter, embed_dataset
    from vpd_amd.augment import CropAugmenter
    from vpd_amd.data import RGB_MEAN_STD
    from vpd_amd.models.rgb import RGBF_EmbeddingModel

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    enc = RGBF_EmbeddingModel(ARCH, EMB_DIM, True, dev, dtype=args.dtype)
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
    del xh

    # ---- 4. u8 frames from the host, views built on the device ----
    aug = CropAugmenter(dev, RGB_MEAN_STD["diving48"], HW, True)
    gc = torch.Generator().manual_seed(2)
    pool = [(torch.randint(0, 256, (FRAMES, HW, HW, 3), generator=gc, dtype=torch.uint8).pin_memory(),
             torch.randint(100, 150, (FRAMES, HW, HW, 2), generator=gc, dtype=torch.uint8).pin_memory()) for _ in range(4)]

    def u8_loader(nbatches, video_of, frame_of):
        f = 0
        for b in range(nbatches):
            idx = torch.arange(f, f + FRAMES)
            rgb, flow = pool[b % len(pool)]
            yield {"video": video_of(idx), "frame": frame_of(idx), "rgb_u8": rgb, "flow_u8": flow}
            f += FRAMES

    embed_dataset(enc, u8_loader(args.warmup, lambda i: torch.zeros_like(i), lambda i: i), 1, augmenter=aug, flip=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    embs = embed_dataset(enc, u8_loader(args.batches, lambda i: i // frames_per_video, lambda i: i % frames_per_video), n_videos,
                         augmenter=aug, flip=True)
    torch.cuda.synchronize()
    dt4 = time.perf_counter() - t0
    assert sum(len(v) for v in embs) == args.batches * FRAMES

    # ---- 5. the whole job at its stated size: real pickles per video ----
    full = None
    if args.crops:
        import shutil
        import numpy as np
        nframes = args.crops // K
        nb = nframes // FRAMES                      # whole batches (the tail batch has its own graph in the product; here: exact multiple)
        nframes = nb * FRAMES
        # Diving48-shaped video lengths: log-normal around 157 frames clipped to [25, 823] (data/sports.cache meta: SURVEY 8d)
        rs = np.random.RandomState(0)
        lens = []
        while sum(lens) < nframes:
            lens.append(int(np.clip(np.round(np.exp(rs.normal(np.log(140.0), 0.5))), 25, 823)))
        lens[-1] -= sum(lens) - nframes
        if lens[-1] <= 0:
            lens.pop()
            lens[-1] += nframes - sum(lens)
        starts = np.concatenate([[0], np.cumsum(lens)])
        vid_of_frame = torch.from_numpy(np.repeat(np.arange(len(lens)), lens))
        frame_in_vid = torch.from_numpy(np.concatenate([np.arange(l) for l in lens]))
        videos = ["video%05d" % i for i in range(len(lens))]
        shutil.rmtree(args.out_dir, ignore_errors=True)
        writer = StreamingWriter(args.out_dir, videos, lens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        embed_dataset(enc, u8_loader(nb, lambda i: vid_of_frame[i], lambda i: frame_in_vid[i]), len(videos), writer=writer,
                      augmenter=aug, flip=True)
        torch.cuda.synchronize()
        dt5 = time.perf_counter() - t0
        files = os.listdir(args.out_dir)
        nbytes = sum(os.path.getsize(os.path.join(args.out_dir, f)) for f in files)
        # spot-check: a pickle in the reference's format, sorted, complete
        from vpd_amd.io import load_pickle
        chk = load_pickle(os.path.join(args.out_dir, videos[len(videos) // 2] + ".emb.pkl"))
        ok = (len(chk) == lens[len(videos) // 2] and [t[0] for t in chk] == list(range(len(chk))) and
              chk[0][1].shape == (K, EMB_DIM) and chk[0][1].dtype.name == "float32" and chk[0][2] == {})
        full = {"crops": nframes * K, "frames": nframes, "videos": len(videos), "mean_frames_per_video": nframes / len(videos),
                "seconds": dt5, "crops_per_s": nframes * K / dt5, "pickles_written": len(files), "pickle_MB": nbytes / 1e6,
                "pickle_format_ok": bool(ok), "out_dir": args.out_dir,
                "note": "u8 frames from pinned host memory (pool of 4 batches re-used), views on the device, hipGraph forward, "
                        "D2H of embeddings, tuple assembly and one pickle per video written as the video completes"}
        shutil.rmtree(args.out_dir, ignore_errors=True)
    res = {"metric": "frame-crops/sec (VPD student apply, eval forward)", "unit": "crops/s",
           "workload": "configs[4]-shaped: %d batches of %d frames x %d views, ResNet-34 5x128x128, D=%d, bf16, hipGraph"
                       % (args.batches, FRAMES, K, EMB_DIM),
           "forward_resident": crops / dt1, "loop_resident": crops / dt2, "loop_host_fp32": crops / dt3,
           "loop_host_u8": crops / dt4, "full_run": full,
           "ms_per_batch_forward": 1e3 * dt1 / args.batches,
           "forward_tflops": crops / dt1 * FWD_FLOP_PER_CROP / 1e12,
           "h2d_GBps_in_loop": crops * C_IN * HW * HW * 4 / dt3 / 1e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
