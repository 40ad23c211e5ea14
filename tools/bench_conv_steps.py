#!/usr/bin/env python3
"""Per-K-step cost of the warp-specialised 3x3 conv kernels: the same output tiling with Ci = 256 / 512 / 1024 / 2048
(36 ... 288 K-steps per block) separates the fixed cost of a launch (prologue, epilogue, dispatch) from the steady state.
Usage: python tools/bench_conv_steps.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpd_amd._lib import check, lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = lib()
ptr = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
tap = lambda *v: (C.c_int * 9)(*v)


def timeit(fn, iters=30):
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


for co, hw in ((128, 16), (256, 8), (512, 4)):
    prev = None
    for ci in (co, 2 * co, 4 * co, 8 * co):
        x = torch.randn(B * (hw + 2) * (hw + 2) * ci, device="cuda").to(torch.bfloat16)
        w = (torch.randn(9 * co * ci, device="cuda") * 0.05).to(torch.bfloat16)
        y = torch.zeros(B * hw * hw * co, device="cuda", dtype=torch.bfloat16)
        stats = torch.zeros(64 * 2 * co, dtype=torch.float64, device="cuda")
        taps = tap(3, 3, 0, 1, 0, 1, 0, 3, 1)

        def fwd():
            check(L.vpd_op_conv2d(ptr(x), ptr(w), ptr(y), ptr(stats), B, hw + 2, hw + 2, ci, hw, hw, co, 0, hw, hw, 1, 0,
                                  0, 1, ci, co, taps, 0, st()), "conv")
        t = timeit(fwd)
        steps = ci // 64 * 9
        flops = 2.0 * B * hw * hw * co * ci * 9
        extra = "" if prev is None else "  marginal %.0f ns/step" % ((t - prev[0]) / (steps - prev[1]) * 1e3)
        print("Co %4d %2dx%-2d Ci %5d  %4d steps  %7.1f us  %5.0f TF/s%s" % (co, hw, hw, ci, steps, t, flops / t / 1e6, extra))
        prev = (t, steps)
