"""ctypes binding of libtsg_hip.so (C-ABI in include/tsg_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call
returns non-zero, the caller gets an exception.  The product path never routes
around the HIP kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtsg_hip.so")

F32, BF16 = 0, 1
NCHW, NHWC = 0, 1
I64, U8 = 0, 1

_ERR = {-1: "unsupported dtype", -2: "unsupported layout", -3: "bad shape",
        -4: "misaligned pointer", -5: "null pointer", -6: "workspace too small",
        -7: "librccl.so could not be loaded"}


class TsgError(RuntimeError):
    pass


class OhemPlan(C.Structure):
    _fields_ = [("P", C.c_int64), ("C", C.c_int), ("grid", C.c_int), ("levels", C.c_int),
                ("shift", C.c_int * 3), ("bins", C.c_int * 3), ("thresh_bits", C.c_uint32),
                ("ws_bytes", C.c_size_t)]


_p, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); kept in lock-step with include/tsg_hip.h
# (tests/test_abi.py parses the header and checks every symbol is exported).
_PROTOS = {
    "tsg_version": (_i, []),
    "tsg_bn_num_partials": (_i, [_i, _i64, _i64, _i64]),
    "tsg_bn_partial_ws_bytes": (_sz, [_i, _i64, _i64, _i64]),
    "tsg_bn_stats": (_i, [_p, _i, _i, _i64, _i64, _i64, _p, _ip, _p]),
    "tsg_bn_collapse": (_i, [_p, _i, _i64, _p, _p]),
    "tsg_bn_collapse_count": (_i, [_p, _i, _i64, _p, _i64, _p]),
    "tsg_bn_finalize": (_i, [_p, _i, _i64, _d, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "tsg_bn_affine": (_i, [_p, _p, _p, _p, _i64, _p, _p]),
    "tsg_bn_apply_fwd": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _i64, _p, _i, _p]),
    "tsg_bn_bwd_reduce": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _i64, _p, _i, _p, _ip, _p]),
    "tsg_bn_bwd_coeffs": (_i, [_p, _i, _i64, _d, _p, _i, _p, _p, _p, _p, _p, _p]),
    "tsg_bn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _p, _i, _p]),
    "tsg_bn_maskbits_supported": (_i, [_i, _i, _i64, _i64]),
    "tsg_bn_apply_fwd_maskbits": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _p, _p]),
    "tsg_bn_bwd_reduce_maskbits": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _i64, _p, _p, _p, _p]),
    "tsg_bn_bwd_apply_maskbits": (_i, [_p, _p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _p, _p]),
    "tsg_bn_mixed_supported": (_i, [_i, _i64, _i64]),
    "tsg_bn_mixed_num_partials": (_i, [_i64, _i64, _i64]),
    "tsg_bn_apply_fwd_mixed": (_i, [_p, _p, _i, _i64, _i64, _i64, _p, _i, _p]),
    "tsg_bn_bwd_reduce_mixed": (_i, [_p, _p, _i, _i64, _i64, _i64, _p, _i, _p, _ip, _p]),
    "tsg_bn_bwd_apply_mixed": (_i, [_p, _p, _p, _i, _i64, _i64, _i64, _p, _i, _p]),
    "tsg_gap_ws_bytes": (_sz, [_i, _i64, _i64, _i64]),
    "tsg_gap_fwd": (_i, [_p, _p, _i, _i, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_gap_bwd": (_i, [_p, _p, _i, _i, _i64, _i64, _i64, _p]),
    "tsg_adaptive_avgpool_nhwc_ws_bytes": (_sz, [_i, _i64, _i, _i, _i, _i, _i]),
    "tsg_adaptive_avgpool_nhwc_fwd": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "tsg_adaptive_avgpool_nhwc_bwd": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _i, _p]),
    "tsg_cat2_rows": (_i, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "tsg_chanscale_fwd": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _i64, _i, _p]),
    "tsg_chanscale_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _i, _p, _sz, _p]),
    "tsg_chanscale_bwd_ds": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_chanscale_bwd_dx": (_i, [_p, _p, _p, _p, _i, _i, _i64, _i64, _i64, _i, _p]),
    "tsg_maxpool_nhwc_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tsg_maxpool_nhwc_bwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tsg_stem_conv_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64]),
    "tsg_stem_conv_ws_bytes": (_sz, []),
    "tsg_stem_conv_fwd": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_weight_shadow_entry_bytes": (_sz, []),
    "tsg_weight_shadow_refresh": (_i, [_p, _p, _i64, _p]),
    "tsg_stem_conv_wrw_bn": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_stem_conv_stats_partials": (_i, [_i64, _i64, _i64]),
    "tsg_conv3x3_wrw_tr_norm": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_conv3x3_wrw_gen_norm": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _p, _sz, _p]),
    "tsg_conv3x3_c64_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tsg_conv3x3_c64_stats_partials": (_i, [_i64, _i64, _i64]),
    "tsg_conv3x3_c64_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "tsg_conv3x3_c64_s2_stats_partials": (_i, [_i64, _i64, _i64]),
    "tsg_conv3x3_c64_s2_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "tsg_conv3x3_c64_s2_dgrad": (_i, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "tsg_conv3x3_c64_dgrad_bnsums_partials": (_i, [_i64, _i64, _i64]),
    "tsg_conv3x3_c64_dgrad_bnsums": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "tsg_conv3x3_c64_s2_dgrad_partials": (_i, [_i64, _i64, _i64]),
    "tsg_conv3x3_c64_s2_dgrad_bnsums": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "tsg_conv3x3_gen_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tsg_conv3x3_gen_filter_elems": (_i64, [_i, _i]),
    "tsg_conv3x3_gen_tile": (_i, [_i64, _i64, _i64, _i, _i]),
    "tsg_conv3x3_gen_prep_filter": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "tsg_conv3x3_gen_stats_partials": (_i, [_i64, _i64, _i64, _i, _i, _i]),
    "tsg_conv3x3_gen_variant": (_i, [_i64, _i64, _i64, _i, _i, _i, _i]),
    "tsg_conv3x3_s2_dgrad_supported": (_i, [_i, _i, _i]),
    "tsg_conv3x3_s2_dgrad": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p]),
    "tsg_conv3x3_s2_dgrad_subadd": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p]),
    "tsg_conv3x3_gen_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _p]),
    "tsg_stem_conv_fwd_stats": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_bn_relu_pool_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _p, _p]),
    "tsg_bn_relu_pool_bwd_num_partials": (_i, [_i, _i64, _i, _i, _i]),
    "tsg_bn_relu_pool_bwd_reduce": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _p, _p, _p]),
    "tsg_bn_relu_pool_bwd_apply": (_i, [_p, _p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _p, _p]),
    "tsg_stem_conv_wrw": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_stem_conv_stats": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_stem_conv_bn_relu_pool_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_stem_pool_bwd_num_partials": (_i, [_i64, _i64, _i64]),
    "tsg_stem_conv_bn_relu_pool_bwd_reduce": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_stem_conv_wrw_bn_pool": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_confusion_map": (_i, [_p, _i, _p, _i, _i64, _i, _p, _p]),
    "tsg_confusion_logits": (_i, [_p, _i, _p, _i, _i64, _i, _i64, _i, _p, _p]),
    "tsg_sgd_multi_blockmap": (_i64, [_p, _i, _p, _i64]),
    "tsg_sgd_multi_step_dev": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _p, _i64, _f, _p]),
    "tsg_conv3x3_wrw_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tsg_conv3x3_wrw_ws_bytes": (_sz, []),
    "tsg_conv3x3_wrw": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_conv3x3_wrw_tr": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_conv3x3_wrw_gen_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tsg_conv3x3_wrw_gen_ws_bytes": (_sz, [_i64, _i64, _i64, _i, _i, _i]),
    "tsg_conv3x3_wrw_gen": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _p, _sz, _p]),
    "tsg_conv3x3_weight_rot180_t": (_i, [_p, _i, _p, _i, _i, _p]),
    "tsg_multi_copy_f32": (_i, [_p, _p, _p, _i, _p, _i64, _f, _p]),
    "tsg_ohem_make_plan": (_i, [_i64, _i, _i64, _f, C.POINTER(OhemPlan)]),
    "tsg_ohem_fwd": (_i, [_p, _i, _p, _i, _i64, _i, _i64, _i64, _f, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "tsg_ohem_bwd": (_i, [_p, _i, _p, _i, _i64, _i, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "tsg_ohem_up_supported": (_i, [_i, _i, _i, _i, _i, _f]),
    "tsg_ohem_up_fwd": (_i, [_p, _i, _p, _i, _i64, _i, _i, _i, _i, _i, _i64, _f, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "tsg_ohem_up_bwd_ws_bytes": (_sz, [_i64, _i, _i, _i]),
    "tsg_ohem_up_bwd": (_i, [_p, _i, _p, _i, _i64, _i, _i, _i, _i, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "tsg_ohem_target_prob": (_i, [_p, _p, _i, _i64, _i, _i64, _p, _p]),
    "tsg_kth_ws_bytes": (_sz, [_i64]),
    "tsg_kth_value": (_i, [_p, _i64, _i64, _p, _p, _sz, _p]),
    "tsg_focal_ws_bytes": (_sz, [_i64]),
    "tsg_focal_fwd": (_i, [_p, _i, _p, _i, _i64, _i64, _f, _f, _p, _p, _sz, _p]),
    "tsg_focal_bwd": (_i, [_p, _i, _p, _i, _i64, _i64, _f, _f, _p, _p, _p]),
    "tsg_upsample_bilinear_ac_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _p]),
    "tsg_upsample_bilinear_ac_bwd": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _p]),
    "tsg_upsample_bilinear_ac_nhwc_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _p]),
    "tsg_upsample_bilinear_ac_nhwc_bwd": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _i, _p]),
    "tsg_upsample_bilinear_ac_presum_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _p]),
    "tsg_upsample_bilinear_ac_nhwc_presum_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _p]),
    "tsg_upsample_nearest_fwd": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _p]),
    "tsg_psa_ws_bytes": (_sz, [_i, _i, _i64, _i64, _i64, _i64]),
    "tsg_psa_fwd": (_i, [_p, _p, _p, _p, _i, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_psa_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "tsg_augment_max_samples": (_i, []),
    "tsg_augment_crop": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _f, _i, _p, _p, _i, _p]),
    "tsg_resize_bilinear_hp": (_i, [_p, _p, _i, _i64, _i, _i, _i, _i, _i, _p]),
    "tsg_comm_init_library": (_i, [C.c_char_p]),
    "tsg_comm_unique_id_bytes": (_i, []),
    "tsg_comm_get_unique_id": (_i, [_p]),
    "tsg_comm_create": (_i, [_p, _i, _i, _i, C.POINTER(_p)]),
    "tsg_comm_destroy": (_i, [_p]),
    "tsg_comm_rank": (_i, [_p]),
    "tsg_comm_world": (_i, [_p]),
    "tsg_comm_allreduce": (_i, [_p, _p, _i64, _i, _p]),
    "tsg_comm_allgather": (_i, [_p, _p, _p, _i64, _i, _p]),
    "tsg_comm_reduce_scatter": (_i, [_p, _p, _p, _i64, _i, _p]),
    "tsg_comm_broadcast": (_i, [_p, _p, _i64, _i, _i, _p]),
    "tsg_comm_error_string": (C.c_char_p, [_i]),
    "tsg_comm_xgmi_handle_bytes": (_sz, []),
    "tsg_comm_xgmi_export": (_i, [_p, _i64, _p]),
    "tsg_comm_xgmi_attach": (_i, [_p, _p]),
    "tsg_xgmi_small_allreduce": (_i, [_p, _p, _i64, _p]),
    "tsg_sgd_step_dev": (_i, [_p, _p, _p, _i64, _p, _f, _f, _f, _f, _p]),
    "tsg_sgd_step": (_i, [_p, _p, _p, _i64, _f, _f, _f, _f, _i, _p]),
    "tsg_edge_labels_ws_bytes": (_sz, [_i, _i]),
    "tsg_edge_labels": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    "tsg_conv1x1_vec_supported": (_i, [_i, _i, _i]),
    "tsg_conv1x1_vec_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "tsg_conv1x1_vec_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tsg_conv1x1_vec_bnact_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _i, _i, _i, _i, _p]),
    "tsg_conv1x1_vec_bnact_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "tsg_cls_head_supported": (_i, [_i, _i, _i, _i64]),
    "tsg_cls_head_fwd": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _i, _p]),
    "tsg_cls_head_dgrad": (_i, [_p, _p, _p, _i64, _i64, _i, _i, _p]),
    "tsg_cls_head_wgrad_ws_bytes": (_sz, [_i64, _i, _i]),
    "tsg_cls_head_wgrad": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _i, _p, _sz, _p]),
    "tsg_conv2d_f32_exact_fwd": (_i, [_p, _p, _p, _i64] + [_i] * 12 + [_p, _p, _p, _p]),
    "tsg_conv2d_f32_exact_dgrad": (_i, [_p, _p, _p, _i64] + [_i] * 12 + [_p, _p, _p, _p]),
    "tsg_conv2d_f32_exact_wgrad_ws_bytes": (_sz, [_i64] + [_i] * 12),
    "tsg_conv2d_f32_exact_wgrad": (_i, [_p, _p, _p, _i64] + [_i] * 12 + [_p, _p, _p, _p, _sz, _p]),
}

_lib = None


def lib():
    """Return the loaded CDLL; raise if libtsg_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TsgError(
                f"{LIB_PATH} not found: build it with `python -m torchseg_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback.")
        # torch FIRST: its wheel bundles its own libamdhip64.so; libtsg_hip.so is linked against /opt/rocm's.  With torch's
        # runtime already in the process ours resolves to it (same SONAME) and kernels, streams and pointers belong to
        # ONE HIP runtime; loaded the other way round (`python __graft_entry__.py --smoke`: build() before torch) every
        # launch on a torch stream came back hipErrorNoDevice (round 5)
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if handle.tsg_version() < 100:
            raise TsgError("libtsg_hip.so is older than this package")
        _lib = handle
    return _lib


def check(rc, what):
    if rc == 0:
        return
    if rc <= -100:
        raise TsgError(f"{what}: RCCL error {-100 - rc} ({lib().tsg_comm_error_string(rc).decode()})")
    if rc < 0:
        raise TsgError(f"{what}: invalid argument ({_ERR.get(rc, rc)})")
    raise TsgError(f"{what}: hipError_t {rc}")


def ptr(t):
    """Device (or host) address of a tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TsgError(f"unsupported activation dtype {t.dtype} (float32 / bfloat16 only)")


_raw_stream = None


def stream_ptr(t):
    """hipStream_t of torch's current stream on the tensor's device (also right under
    stream contexts and hipGraph capture).  Uses torch's raw-stream getter: this runs
    ~1200 times per training step."""
    global _raw_stream
    if not t.is_cuda:
        raise TsgError("torchseg_amd kernels need tensors on an AMD GPU (got a CPU tensor); "
                       "there is no CPU fallback in the product path")
    if _raw_stream is None:
        import torch
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
            lambda idx: torch.cuda.current_stream(idx).cuda_stream)
    return _raw_stream(t.device.index)
