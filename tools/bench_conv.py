#!/usr/bin/env python3
"""Per-layer microbenchmark of the conv kernels through the C ABI (random bf16 data, HIP events).
Shapes are the ResNet-34 student's 3x3 convs at B crops of 128x128.  Usage: python tools/bench_conv.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpd_amd._lib import check, lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ABL = int(os.environ.get("VPD_ABLATE", "0"))
L = lib()
ptr = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
tap = lambda *v: (C.c_int * 9)(*v)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


LAYERS = [("layer1", 64, 64, 32), ("layer2", 128, 128, 16), ("layer3", 256, 256, 8), ("layer4", 512, 512, 4)]
print("B=%d ablate=%d" % (B, ABL))
for name, ci, co, hw in LAYERS:
    x = torch.randn(B * (hw + 2) * (hw + 2) * ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(9 * co * ci, device="cuda") * 0.05).to(torch.bfloat16)
    y = torch.zeros(B * hw * hw * co, device="cuda", dtype=torch.bfloat16)
    dzp = torch.randn(B * (hw + 2) * (hw + 2) * co, device="cuda").to(torch.bfloat16)
    stats = torch.zeros(64 * 2 * co, dtype=torch.float64, device="cuda")
    dw = torch.zeros(9 * co * ci, device="cuda")
    slab = torch.empty(L.vpd_op_wgrad_slab_bytes() // 4, device="cuda")
    flops = 2.0 * B * hw * hw * co * ci * 9
    taps = tap(3, 3, 0, 1, 0, 1, 0, 3, 1)

    def fwd():
        check(L.vpd_op_conv2d(ptr(x), ptr(w), ptr(y), None if os.environ.get("NOSTATS") else ptr(stats), B, hw + 2, hw + 2, ci, hw, hw, co, 0, hw, hw, 1, 0,
                              0, 1, ci, co, taps, 0, st()), "conv")

    def wgrad():
        check(L.vpd_op_wgrad(ptr(dzp), ptr(x), ptr(dw), B, hw + 2, hw + 2, co, 1, hw + 2, hw + 2, ci, hw, hw, 1, ci, co,
                             taps, ptr(slab), st()), "wgrad")

    t1 = timeit(fwd)
    t2 = timeit(wgrad)
    print("%s  fwd %7.1f us %6.0f TF/s   wgrad %7.1f us %6.0f TF/s" % (name, t1, flops / t1 / 1e6, t2, flops / t2 / 1e6))
