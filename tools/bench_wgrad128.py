#!/usr/bin/env python3
"""Microbenchmark of the grouped 128 x 64 weight-gradient launch (conv_wgrad128_persistent_kernel) at the ResNet-34
shapes of the 256-crop step, through the C ABI op entry point:  layer2 alone (7 convs), layer3 + layer4 in one launch
(11 + 5 convs), layer3 alone, layer4 alone.  Prints microseconds per launch (HIP events over `reps` launches, the slab
sum included) and TFLOP/s.   VPD_LIB_PATH selects the library; VPD_WG2_* environment switches apply."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vpd_amd._lib import lib, check

L = lib()
N = int(os.environ.get("BENCH_N", "256"))
GROUPS = {
    "layer2": [(N, 16, 16, 128, 128, 1)] * 7 + [(N, 16, 16, 128, 64, 2)],
    "layer3+4": [(N, 8, 8, 256, 256, 1)] * 11 + [(N, 4, 4, 512, 512, 1)] * 5 + [(N, 8, 8, 256, 128, 2), (N, 4, 4, 512, 256, 2)],
    "layer3": [(N, 8, 8, 256, 256, 1)] * 11,
    "layer4": [(N, 4, 4, 512, 512, 1)] * 5,
}


def run(name, probs, reps=20):
    g = torch.Generator(device="cuda").manual_seed(1)
    keep, dzs, xs, dws, slabs, dims = [], [], [], [], [], []
    flops = 0.0
    for pr in probs:
        n, h, w, co, ci, st = pr[:6]
        ks = pr[6] if len(pr) > 6 else 3
        x = torch.zeros(n, st * h + 2, st * w + 2, ci, dtype=torch.bfloat16, device="cuda")
        x[:, 1:-1, 1:-1, :] = torch.randn(n, st * h, st * w, ci, generator=g, device="cuda").clamp_min(0).to(torch.bfloat16)      # post-ReLU activations: half zeros, as in the step
        dz = torch.zeros(n, h + 2, w + 2, co, dtype=torch.bfloat16, device="cuda")
        dz[:, 1:-1, 1:-1, :] = torch.randn(n, h, w, co, generator=g, device="cuda").to(torch.bfloat16)
        dw = torch.empty(ks * ks, co, ci, dtype=torch.float32, device="cuda")
        slab = torch.empty(max(int(L.vpd_op_wgrad128_slab_floats(co, ci)), 4), dtype=torch.float32, device="cuda")
        keep += [x, dz, dw, slab]
        dzs.append(dz.data_ptr()); xs.append(x.data_ptr()); dws.append(dw.data_ptr()); slabs.append(slab.data_ptr())
        dims += [n, h, w, co, ci, st, ks]
        flops += 2.0 * n * h * w * co * ci * ks * ks
    k = len(probs)
    arr = lambda v: (C.c_void_p * k)(*v)
    table = torch.empty(int(L.vpd_op_wgrad128_table_bytes()), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (k, arr(dzs), arr(xs), arr(dws), arr(slabs), (C.c_int * (7 * k))(*dims), C.c_void_p(table.data_ptr()), st)
    for _ in range(3):
        check(L.vpd_op_wgrad128_group(*args), "wgrad128")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        check(L.vpd_op_wgrad128_group(*args), "wgrad128")
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print("%-9s %2d convs: %7.1f us per launch  %6.0f TFLOP/s" % (name, k, us, flops / us / 1e6), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or list(GROUPS)
    for name in which:
        run(name, GROUPS[name])
