#!/usr/bin/env python3
"""The u8 apply loop alone (bench.py's `loop_host_u8` leg), for a profiler:  rocprofv3 --kernel-trace --memory-copy-trace
--output-format csv -d <dir> -o p -- python3 tools/apply_loop_probe.py [batches];  tools/apply_loop_probe.py --analyse <dir>/p
prints, per batch, when the GPU worked and where it idled.  VPD_APPLY_PROBE=1 adds host-side phase times."""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def analyse(prefix):
    k = list(csv.DictReader(open(prefix + "_kernel_trace.csv")))
    k.sort(key=lambda r: int(r["Start_Timestamp"]))
    cp = []
    if os.path.exists(prefix + "_memory_copy_trace.csv"):
        cp = list(csv.DictReader(open(prefix + "_memory_copy_trace.csv")))
    starts = [i for i, r in enumerate(k) if "aug_views_kernel" in r["Kernel_Name"]]
    t0 = int(k[starts[0]]["Start_Timestamp"])
    print("batch  views_start  fwd_first  fwd_last_end  busy_us  idle_before_us  (memcpy in the idle window: dir bytes us)")
    prev_end = None
    for a, b in zip(starts, starts[1:] + [len(k)]):
        seg = k[a:b]
        s, e = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
        idle = (s - prev_end) / 1e3 if prev_end else 0.0
        inwin = []
        if prev_end:
            for c in cp:
                cs, ce = int(c["Start_Timestamp"]), int(c["End_Timestamp"])
                if ce > prev_end - 50000 and cs < s + 50000:
                    inwin.append("%s %s %.0fus@%+.0f" % (c.get("Direction", "?")[-12:], c.get("Bytes", "?"), (ce - cs) / 1e3, (cs - prev_end) / 1e3))
        print("%3d %10.1f %10.1f %12.1f %9.1f %10.1f   %s" % (starts.index(a), (s - t0) / 1e3, (int(seg[1]["Start_Timestamp"]) - t0) / 1e3 if len(seg) > 1 else 0,
                                                            (e - t0) / 1e3, busy, idle, "; ".join(inwin)))
        prev_end = e


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        return analyse(sys.argv[2])
    import torch
    from vpd_amd.apply import embed_dataset
    from vpd_amd.augment import CropAugmenter
    from vpd_amd.data import RGB_MEAN_STD
    from vpd_amd.models.rgb import RGBF_EmbeddingModel
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    frames, hw = 500, 128
    dev = torch.device("cuda", 0)
    enc = RGBF_EmbeddingModel("resnet34", 128, True, dev)
    enc.reset_parameters(seed=0)
    enc.eval()
    aug = CropAugmenter(dev, RGB_MEAN_STD["diving48"], hw, True)
    gc = torch.Generator().manual_seed(2)
    pool = [(torch.randint(0, 256, (frames, hw, hw, 3), generator=gc, dtype=torch.uint8).pin_memory(),
             torch.randint(100, 150, (frames, hw, hw, 2), generator=gc, dtype=torch.uint8).pin_memory()) for _ in range(2)]

    def loader(nb):
        for b in range(nb):
            idx = torch.arange(b * frames, (b + 1) * frames)
            yield {"video": idx // 157, "frame": idx % 157, "rgb_u8": pool[b % 2][0], "flow_u8": pool[b % 2][1]}
    embed_dataset(enc, loader(3), 3 * frames // 157 + 1, augmenter=aug, flip=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    embs = embed_dataset(enc, loader(batches), batches * frames // 157 + 1, augmenter=aug, flip=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("u8 loop: %d batches, %.3f ms per batch, %.0f crops/s" % (batches, 1e3 * dt / batches, batches * frames * 2 / dt))
    assert sum(len(v) for v in embs) == batches * frames


if __name__ == "__main__":
    main()
