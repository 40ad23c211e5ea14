"""CPU-side checks of the C-ABI boundary: the shared library loads, exports
every symbol include/vpd_hip.h declares, and its host-only plan functions
describe the reference's state_dict layout (no GPU compute is called)."""
import ctypes as C
import os
import re

import pytest

from oracle import vpd_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(REPO, "include", "vpd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vpd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vpd_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES.keys()) == names
    h = _lib.lib()
    for n in names:
        assert getattr(h, n) is not None
    assert h.vpd_abi_version() == _lib.ABI_VERSION
    assert h.vpd_elem_dtype() == b"bf16"


def test_fp16_library_exports_the_same_abi():
    """libvpdhip_f16.so = the same sources with fp16 elements: same symbols, same ABI version, its own element type; the loss
    scale (fp16 training) is validated on the host."""
    from vpd_amd import _lib
    h = _lib.lib("fp16")
    for n in header_functions():
        assert getattr(h, n) is not None
    assert h.vpd_abi_version() == _lib.ABI_VERSION and h.vpd_elem_dtype() == b"fp16"
    p = C.c_void_p()
    _lib.check(h.vpd_plan_create(b"resnet34", 5, 128, 128, 128, 0, 4, 1, C.byref(p)), "create", "fp16")
    assert h.vpd_plan_param_numel(p) == 21356608      # (host-only call on the fp16 plan: ResNet-34, 5 channels, D = 128, no motion head)
    assert h.vpd_plan_set_loss_scale(p, 256.0) == 0
    for bad in (0.0, -1.0, float("inf"), float("nan")):
        assert h.vpd_plan_set_loss_scale(p, bad) != 0 and b"loss scale" in h.vpd_last_error()
    h.vpd_plan_destroy(p)
    with pytest.raises(_lib.VpdHipError):
        _lib.lib("fp32")


@pytest.mark.parametrize("arch,c_in,D,motion", [("resnet34", 5, 128, 1), ("resnet18", 3, 32, 0)])
def test_plan_layout_matches_reference_schema(arch, c_in, D, motion):
    from vpd_amd._lib import check, lib
    L = lib()
    h = C.c_void_p()
    check(L.vpd_plan_create(arch.encode(), c_in, 128, 128, D, motion, 4, 1, C.byref(h)), "create")
    sch = O.encoder_schema(arch, c_in, D)
    names = O.trainable_keys(sch)
    shapes = [tuple(sch[k][0]) for k in names]
    if motion:
        ds = O.decoder_schema(D)
        names += list(ds.keys())
        shapes += [tuple(v[0]) for v in ds.values()]
    assert L.vpd_plan_num_tensors(h) == len(names)
    kind, dec, off, numel, ndim = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong(), C.c_int()
    dims = (C.c_int * 4)()
    expect_off = 0
    for i, shp in enumerate(shapes):
        check(L.vpd_plan_tensor_info(h, i, C.byref(kind), C.byref(dec), C.byref(off), C.byref(numel), C.byref(ndim), dims), "info")
        assert tuple(dims[k] for k in range(ndim.value)) == shp, names[i]
        assert off.value == expect_off
        expect_off += numel.value
    total = expect_off
    if arch == "resnet34" and c_in == 5 and D == 128:
        assert total == 21356608 + 66048            # SURVEY.md 2.3 parameter counts
    assert L.vpd_plan_param_numel(h) == (total + 3) // 4 * 4
    # buckets tile the flat buffer exactly, in reverse stage order
    rng = []
    o, m = C.c_longlong(), C.c_longlong()
    for b in range(L.vpd_plan_num_buckets(h)):
        check(L.vpd_plan_bucket_range(h, b, C.byref(o), C.byref(m)), "bucket")
        rng.append((o.value, m.value))
    rng_sorted = sorted(rng)
    assert rng_sorted[0][0] == 0 and sum(m for _, m in rng) == total
    for (o1, m1), (o2, _) in zip(rng_sorted, rng_sorted[1:]):
        assert o1 + m1 == o2
    assert rng[0][0] > rng[1][0] > rng[2][0] > rng[3][0] == 0
    nbn = sum(1 for k, (_, kind_) in sch.items() if kind_ == "bn_rm")
    assert L.vpd_plan_num_bn(h) == nbn
    assert L.vpd_plan_workspace_bytes(h) > 0
    L.vpd_plan_destroy(h)


def test_plan_rejects_bad_arguments():
    from vpd_amd._lib import lib
    L = lib()
    h = C.c_void_p()
    assert L.vpd_plan_create(b"effnet-b0", 5, 128, 128, 128, 0, 4, 1, C.byref(h)) != 0      # EfficientNet: out of scope
    assert b"unsupported arch" in L.vpd_last_error()
    for arch in (b"resnet50", b"wide_resnet101_2"):          # Bottleneck archs plan without a GPU too
        assert L.vpd_plan_create(arch, 5, 128, 128, 128, 0, 4, 1, C.byref(h)) == 0
        assert L.vpd_plan_param_numel(h) > 20000000
        L.vpd_plan_destroy(h)
    assert L.vpd_plan_create(b"resnet34", 9, 128, 128, 128, 0, 4, 1, C.byref(h)) != 0
    assert L.vpd_plan_create(b"resnet34", 5, 127, 128, 128, 0, 4, 1, C.byref(h)) != 0


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vpd_amd.engine import StudentEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        StudentEngine("resnet18", 5, 32)


def test_product_does_not_import_oracle():
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c",
                          "import sys; import vpd_amd, vpd_amd.engine, vpd_amd.trainer, vpd_amd.models.rgb, vpd_amd.ddp;"
                          "print(any(m.startswith('oracle') for m in sys.modules))"],
                         cwd=REPO, capture_output=True, text=True)
    assert out.stdout.strip() == "False", out.stdout + out.stderr
