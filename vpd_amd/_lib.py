"""ctypes binding of libvpdhip.so (C ABI in include/vpd_hip.h).

The library is the product: there is no PyTorch / CPU fallback for the hot
path.  ``lib()`` raises if the shared object is missing or lacks a symbol.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VPD_LIB_PATH: A/B another build of the same ABI on the same GPU (devices differ by several % in clocks)
LIB_PATH = os.environ.get("VPD_LIB_PATH") or os.path.join(_HERE, "libvpdhip.so")
# the same sources built with fp16 elements (vpd_amd/csrc/Makefile, common.h "Element type"): inference only
LIB_PATH_F16 = os.environ.get("VPD_LIB_PATH_F16") or os.path.join(_HERE, "libvpdhip_f16.so")
ABI_VERSION = 2

c_int_p = C.POINTER(C.c_int)
c_ll_p = C.POINTER(C.c_longlong)
vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/vpd_hip.h one to one
SIGNATURES = {
    "vpd_last_error": (C.c_char_p, []),
    "vpd_abi_version": (C.c_int, []),
    "vpd_elem_dtype": (C.c_char_p, []),
    "vpd_plan_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(vp)]),
    "vpd_plan_destroy": (None, [vp]),
    "vpd_plan_num_tensors": (C.c_int, [vp]),
    "vpd_plan_tensor_info": (C.c_int, [vp, C.c_int, c_int_p, c_int_p, c_ll_p, c_ll_p, c_int_p, c_int_p]),
    "vpd_plan_param_numel": (C.c_longlong, [vp]),
    "vpd_plan_num_bn": (C.c_int, [vp]),
    "vpd_plan_bn_info": (C.c_int, [vp, C.c_int, c_int_p, c_ll_p, c_ll_p]),
    "vpd_plan_bn_numel": (C.c_longlong, [vp]),
    "vpd_plan_num_buckets": (C.c_int, [vp]),
    "vpd_plan_bucket_range": (C.c_int, [vp, C.c_int, c_ll_p, c_ll_p]),
    "vpd_plan_workspace_bytes": (C.c_size_t, [vp]),
    "vpd_plan_init_workspace": (C.c_int, [vp, vp, vp]),
    "vpd_pack_weights": (C.c_int, [vp, vp, vp, vp, vp]),
    "vpd_forward_eval": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]),
    "vpd_forward_train": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]),
    "vpd_backward": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(vp), vp, vp]),
    "vpd_adamw_step": (C.c_int, [vp, vp, vp, vp, C.c_longlong, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_int, vp]),
    "vpd_plan_adamw_step": (C.c_int, [vp, vp, vp, vp, vp, C.c_longlong, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_int, vp, vp]),
    "vpd_augment_crops": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                    C.c_float, vp, vp, vp]),
    "vpd_plan_stage_crops": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                       C.c_float, vp, vp, vp]),
    "vpd_plan_stage_views": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), vp, vp]),
    "vpd_graph_capture_eval": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp]),
    "vpd_graph_launch_eval": (C.c_int, [vp, C.c_int, vp]),
    "vpd_plan_sync_errors": (C.c_int, [vp, vp, vp, C.POINTER(C.c_uint)]),
    "vpd_plan_set_lazy_grads": (C.c_int, [vp, C.c_int]),
    "vpd_plan_set_loss_scale": (C.c_int, [vp, C.c_float]),
    "vpd_plan_bucket_scratch_range": (C.c_int, [vp, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "vpd_plan_grads_pending": (C.c_int, [vp]),
    "vpd_plan_materialize_grads": (C.c_int, [vp, vp, vp, vp]),
    "vpd_plan_set_timing": (C.c_int, [vp, C.c_int]),
    "vpd_plan_read_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.c_int]),
    "vpd_op_conv2d": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 16 + [c_int_p, C.c_int, vp]),
    "vpd_op_conv_bm": (C.c_int, [C.c_int, C.c_int]),
    "vpd_op_conv2d_ep": (C.c_int, [vp, vp, vp] + [C.c_int] * 13 + [c_int_p, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "vpd_op_conv2d_bnsums": (C.c_int, [vp] * 6 + [C.c_int] * 8 + [c_int_p, C.c_int, vp]),
    "vpd_op_bn_forward": (C.c_int, [vp] * 13 + [C.c_int] * 5 + [C.c_float, C.c_float, vp]),
    "vpd_op_bn_backward_apply": (C.c_int, [vp] * 10 + [C.c_int] * 4 + [vp]),
    "vpd_op_wgrad": (C.c_int, [vp, vp, vp] + [C.c_int] * 13 + [c_int_p, vp, vp]),
    "vpd_op_wgrad_slab_bytes": (C.c_size_t, []),
    "vpd_op_tr_read_probe": (C.c_int, [vp, vp, vp]),
    "vpd_op_wgrad128_table_bytes": (C.c_size_t, []),
    "vpd_op_wgrad128_slab_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "vpd_op_wgrad128_group": (C.c_int, [C.c_int, vp, vp, vp, vp, c_int_p, vp, vp]),
    "vpd_op_wgrad128_schedule": (C.c_int, [C.c_int, c_int_p, C.c_int, c_int_p, c_int_p, c_int_p, C.c_int,
                                           C.POINTER(C.c_double)]),
}

_libs = {}


class VpdHipError(RuntimeError):
    pass


def lib(dtype="bf16"):
    """Load libvpdhip.so (dtype "bf16": training and inference) or libvpdhip_f16.so ("fp16": inference), once each.
    Raises -- never falls back -- when the library is absent, lacks a symbol or was built for another element type."""
    if dtype in _libs:
        return _libs[dtype]
    if dtype not in ("bf16", "fp16"):
        raise VpdHipError("unknown element type %r (bf16 | fp16)" % (dtype,))
    path = LIB_PATH if dtype == "bf16" else LIB_PATH_F16
    name = os.path.basename(path)
    if not os.path.isfile(path):
        raise VpdHipError(
            "%s not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vpd_amd/csrc`.  vpd_amd has no PyTorch/CPU fallback." % (name, path))
    h = C.CDLL(path)
    # VPD_LIB_PATH (same-box A/B against an OLDER build of the library, tools/build_head_lib.sh): the operator-level test entry
    # points that build does not have yet are skipped; the in-tree library must export every declared symbol
    ab_build = "VPD_LIB_PATH" in os.environ and dtype == "bf16"
    for sym, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, sym)
        except AttributeError as e:
            # (an OLDER round's library, same-box A/B: the entry points added since are optional there -- engine.py asks hasattr)
            if ab_build and (sym.startswith("vpd_op_") or sym in ("vpd_elem_dtype", "vpd_plan_set_loss_scale")):
                continue
            raise VpdHipError("%s lacks symbol %s declared in include/vpd_hip.h" % (name, sym)) from e
        fn.restype = res
        fn.argtypes = args
    # (VPD_LIB_ALLOW_ABI=<n>: a same-box A/B against round n's library, whose train-step entry points have the same signatures)
    if h.vpd_abi_version() != ABI_VERSION and not (ab_build and os.environ.get("VPD_LIB_ALLOW_ABI") == str(h.vpd_abi_version())):
        raise VpdHipError("%s ABI version %d != expected %d" % (name, h.vpd_abi_version(), ABI_VERSION))
    if hasattr(h, "vpd_elem_dtype") and h.vpd_elem_dtype().decode() != dtype:
        raise VpdHipError("%s was built with %s elements, %s asked for" % (path, h.vpd_elem_dtype().decode(), dtype))
    _libs[dtype] = h
    return h


def check(rc, what="", dtype="bf16"):
    if rc != 0:
        msg = lib(dtype).vpd_last_error()
        raise VpdHipError("%s failed: %s" % (what or "libvpdhip call", msg.decode() if msg else "unknown error"))
