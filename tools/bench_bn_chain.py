#!/usr/bin/env python3
"""conv -> BatchNorm-forward -> conv -> ... on one stage's shapes through the C ABI operators (the launches of the plan), timed as a chain:
does the XCD-affine block order of the BatchNorm launch (VPD_BN_XCD_FORCE=<tile pixels>, -DVPD_ENABLE_ABLATE build) shorten the pair?
Usage: python tools/bench_bn_chain.py [B]"""
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
for name, c, hw in (("layer2", 128, 16), ("layer3", 256, 8), ("layer4", 512, 4)):
    a = [torch.zeros(B * (hw + 2) * (hw + 2) * c, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    a[0].view(B, hw + 2, hw + 2, c)[:, 1:-1, 1:-1] = torch.randn(B, hw, hw, c, device="cuda").to(torch.bfloat16)
    w = (torch.randn(9 * c * c, device="cuda") * 0.02).to(torch.bfloat16)
    z = torch.zeros(B * hw * hw * c, device="cuda", dtype=torch.bfloat16)
    rows = torch.zeros(4 * 2 * c, dtype=torch.float64, device="cuda")
    gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    coef = [torch.zeros(c, device="cuda") for _ in range(4)]
    mask = torch.zeros(B * hw * hw * c // 8, dtype=torch.uint8, device="cuda")
    taps = tap(3, 3, 0, 1, 0, 1, 0, 3, 1)

    def pair(i):
        rows.zero_()
        check(L.vpd_op_conv2d(ptr(a[i & 1]), ptr(w), ptr(z), ptr(rows), B, hw + 2, hw + 2, c, hw, hw, c, 0, hw, hw, 1, 0, 0, 1, c, c, taps, 0, st()), "conv")
        check(L.vpd_op_bn_forward(ptr(z), ptr(rows), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), *[ptr(t) for t in coef], None, ptr(a[(i + 1) & 1]),
                                  ptr(mask), B, hw, hw, c, 1, C.c_float(0.1), C.c_float(1e-5), st()), "bn")

    for i in range(6):
        pair(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for i in range(n):
        pair(i)
    e1.record()
    torch.cuda.synchronize()
    print("%s B=%d force=%s: %.2f us per (zero + conv + BatchNorm) triple; finite %s" % (name, B, os.environ.get("VPD_BN_XCD_FORCE", "-"), e0.elapsed_time(e1) / n * 1e3,
                                                                                       bool(torch.isfinite(a[0].float()).all())))
