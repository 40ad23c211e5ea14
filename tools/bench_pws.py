#!/usr/bin/env python3
"""Forward 3x3 stride-1 convolutions of layer2 / layer3 / layer4 (train epilogue: dense store + statistics) through the C ABI,
one line per run: `python tools/bench_pws.py B [label]`.  Kernel choice by environment (VPD_PWS, VPD_PWS_VAR, VPD_ABLATE,
VPD_LIB_PATH), read once per process by the library -- run one process per configuration, alternating."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpd_amd._lib import check, lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
label = sys.argv[2] if len(sys.argv) > 2 else ""
L = lib()
ptr = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
tap = lambda *v: (C.c_int * 9)(*v)


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
    return a.elapsed_time(b) / iters * 1e3   # us


out = []
LAYERS = (("l1", 64, 64, 32), ("l2", 128, 128, 16), ("l3", 256, 256, 8), ("l4", 512, 512, 4))
if os.environ.get("BENCH_PWS_LAYERS"):
    LAYERS = tuple(t for t in LAYERS if t[0] in os.environ["BENCH_PWS_LAYERS"].split(","))
for name, ci, co, hw in LAYERS:
    x = torch.randn(B * (hw + 2) * (hw + 2) * ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(9 * co * ci, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.zeros(B * hw * hw * co, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(64 * 2 * co, dtype=torch.float64, device="cuda")
    flops = 2.0 * B * hw * hw * co * ci * 9
    taps = tap(3, 3, 0, 1, 0, 1, 0, 3, 1)

    def fwd():
        check(L.vpd_op_conv2d(ptr(x), ptr(w), ptr(y), ptr(stats), B, hw + 2, hw + 2, ci, hw, hw, co, 0, hw, hw, 1, 0, 0, 1, ci, co,
                              taps, 0, st()), "conv")
    t = timeit(fwd)
    out.append("%s %6.1f us %4.0f TF" % (name, t, flops / t / 1e6))
print("B=%-4d %-22s %s" % (B, label, " | ".join(out)))
