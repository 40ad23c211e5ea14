#!/usr/bin/env python3
"""Algorithmic bandwidth of every fused BatchNorm launch of one ResNet-34 train step (256 crops of 128 x 128), from
tools/step_timeline.py's output:   tools/bn_bandwidth.py <step_timeline.txt> [crops]
The launches are identified by their order in the step (forward: bn1, bn2 of every BasicBlock; backward: the reverse).  Bytes per
element: forward = z read (2) + padded activation written (2 x (H+2)(W+2)/HW) + ReLU bit map (1/8) [+ residual read (2)]; backward-
apply = dy (2) + z (2) + bit map (1/8) + padded dz written [pair launch of a down-sampling block: + z2 (2) + dz2]."""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
stages = [(64, 32, 3), (128, 16, 4), (256, 8, 6), (512, 4, 3)]
fwd, bwd = [], []
for s, (C, H, nb) in enumerate(stages):
    pad = (H + 2) * (H + 2) / float(H * H)
    elems = N * H * H * C
    for b in range(nb):
        fwd.append(("layer%d.%d.bn1" % (s + 1, b), elems * (2 + 2 * pad + 0.125)))
        fwd.append(("layer%d.%d.bn2%s" % (s + 1, b, "+ds" if (b == 0 and s > 0) else ""), elems * (2 + 2 * pad + 0.125 + 2)))
for s in range(3, -1, -1):
    C, H, nb = stages[s]
    pad = (H + 2) * (H + 2) / float(H * H)
    elems = N * H * H * C
    for b in range(nb - 1, -1, -1):
        pair = b == 0 and s > 0
        bwd.append(("layer%d.%d.bn2%s" % (s + 1, b, " (pair launch)" if pair else ""), elems * ((6.125 + 2 * pad) if pair else (4.125 + 2 * pad))))
        bwd.append(("layer%d.%d.bn1" % (s + 1, b), elems * (4.125 + 2 * pad)))
dur = lambda key: [float(re.search(r"gap\s+([0-9.]+) us", l).group(1)) for l in lines if key in l]
f_us = dur("bn_fwd_fused_kernel")
b_us = [float(re.search(r"gap\s+([0-9.]+) us", l).group(1)) for l in lines if "bn_bwd_apply_fused_kernel" in l or "bn_bwd_fused_kernel<3" in l]
print("# forward: %d launches in the step, %d expected; backward: %d launches, %d expected (the last block's bn2 runs on the grid-barrier kernel)"
      % (len(f_us), len(fwd), len(b_us), len(bwd)))
print("%-34s %9s %8s %7s" % ("launch", "MB", "us", "TB/s"))
for (name, by), us in zip(fwd, f_us):
    print("fwd %-30s %9.1f %8.1f %7.2f" % (name, by / 1e6, us, by / us / 1e6))
for (name, by), us in zip(bwd, b_us):
    print("bwd %-30s %9.1f %8.1f %7.2f" % (name, by / 1e6, us, by / us / 1e6))
