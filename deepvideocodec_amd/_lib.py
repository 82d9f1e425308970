"""ctypes binding of libdvc_hip.so (declarations: include/dvc_hip.h).

There is deliberately no fallback: if the shared library is missing or an entry
point fails, the caller gets an exception -- never a silent CPU path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVC_HIP_LIB overrides the in-tree library (kernel experiments); there is still no fallback.
LIB_PATH = os.environ.get("DVC_HIP_LIB") or os.path.join(_HERE, "lib", "libdvc_hip.so")

DVC_MAX_SRC = 4
EB_PARAMS_PER_CHANNEL = 58


class DvcError(RuntimeError):
    pass


class DvcSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("chan_stride", C.c_int64),
                ("channels", C.c_int32), ("_pad", C.c_int32)]


class DvcDst(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_int64), ("chan_stride", C.c_int64)]


class DvcConv2d(C.Structure):
    _fields_ = [
        ("src", DvcSrc * DVC_MAX_SRC),
        ("n_src", C.c_int32),
        ("batch", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("weight_packed", C.c_void_p),
        ("bias", C.c_void_p),
        ("out_channels", C.c_int32),
        ("ksize", C.c_int32),
        ("stride", C.c_int32),
        ("in_act", C.c_int32), ("out_act", C.c_int32),
        ("in_slope", C.c_float), ("out_slope", C.c_float),
        ("pixel_shuffle", C.c_int32),
        ("out", DvcDst),
        ("residual", C.c_void_p),
        ("res_batch_stride", C.c_int64), ("res_chan_stride", C.c_int64),
        ("res_scale", C.c_void_p),
        ("residual2", C.c_void_p),
        ("res2_batch_stride", C.c_int64), ("res2_chan_stride", C.c_int64),
        ("precision", C.c_int32), ("_pad", C.c_int32),
        ("plane_sums", C.c_void_p),
    ]


_P = C.c_void_p
_SRC = C.POINTER(DvcSrc)
_DST = C.POINTER(DvcDst)
_i32, _i64, _f32 = C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every symbol include/dvc_hip.h declares.
SIGNATURES = {
    "dvc_conv2d": (C.c_int, [C.POINTER(DvcConv2d), _P]),
    "dvc_conv2d_plane_sum_slots": (_i32, [C.POINTER(DvcConv2d)]),
    "dvc_conv2d_packed_weight_floats": (_i64, [_i32, _i32, _i32, _i32]),
    "dvc_conv2d_pack_weight": (C.c_int, [_P, _i32, _i32, _i32, _i32, _P, _P]),
    "dvc_conv2d_pack_weight_dgrad": (C.c_int, [_P, _i32, _i32, _i32, _P, _P]),
    "dvc_conv2d_reduced_packed_weight_floats": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "dvc_conv2d_reduced_pack_weight": (C.c_int, [_P, _i32, _i32, _i32, _i32, _i32, _P, _P]),
    "dvc_flow_warp": (C.c_int, [_SRC, _SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_avg_pool2": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _f32, _P]),
    "dvc_max_pool2": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_bilinear_up2": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _f32, _P]),
    "dvc_spynet_level_input": (C.c_int, [_SRC, _SRC, _SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_se_scratch_floats": (_i64, [_i32, _i32]),
    "dvc_se_gate": (C.c_int, [_SRC, _i32, _i32, _i32, _P, _P, _i32, _P, _P, _P]),
    "dvc_se_gate_from_sums": (C.c_int, [_P, _i32, _i32, _i32, _i32, _i32, _P, _P, _i32, _P, _P, _P]),
    "dvc_se_gate_static": (C.c_int, [_SRC, _i32, _i32, _i32, _P, _P, _i32, _P, _P]),
    "dvc_copy": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_clamp": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _f32, _f32, _P]),
    "dvc_eb_forward": (C.c_int, [_SRC, _P, _P, _P, _DST, _DST, _P, _i32, _i32, _i32, _P]),
    "dvc_eb_dequantize": (C.c_int, [_P, _P, _DST, _i32, _i32, _i32, _i32, _P]),
    "dvc_dual_prior_pass1": (C.c_int, [_SRC, _SRC, _SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_process_with_mask": (C.c_int, [_SRC, _SRC, _SRC, _P, _DST, _DST, _DST, _DST, _i32, _i32, _i32, _P]),
    "dvc_dual_prior_pass2": (C.c_int, [_SRC, _SRC, _SRC, _SRC, _P, _DST, _DST, _DST, _DST, _P, _P, _P, _i32,
                                        _i32, _i32, _i32, _P]),
    "dvc_dual_prior_dec_indexes": (C.c_int, [_SRC, _i32, _P, _P, _i32, _i32, _i32, _i32, _i32, _P]),
    "dvc_dual_prior_dec_apply": (C.c_int, [_P, _SRC, _i32, _SRC, _DST, _i32, _i32, _i32, _i32, _P]),
    "dvc_gaussian_conditional": (C.c_int, [_SRC, _SRC, _SRC, _P, _DST, _DST, _i32, _i32, _i32, _P]),
    "dvc_build_indexes": (C.c_int, [_SRC, _P, _P, _i32, _i32, _i32, _i32, _P]),
    "dvc_bits_scratch_floats": (_i64, [_i32]),
    "dvc_likelihood_bits": (C.c_int, [_SRC, _i32, _i32, _i32, _P, _P, _P]),
    "dvc_sq_err_sum": (C.c_int, [_SRC, _SRC, _i32, _i32, _i32, _P, _P, _P]),
    "dvc_conv2d_wgrad_scratch_floats": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "dvc_conv2d_wgrad": (C.c_int, [_SRC, _i32, _i32, _i32, _i32, _i32, _f32, _P, _i32, _i32, _i32, _P, _P, _P, _P]),
    "dvc_act_grad": (C.c_int, [_SRC, _SRC, _f32, _i32, _P, _i32, _i32, _i32, _i32, _P]),
    "dvc_mul_act_grad": (C.c_int, [_SRC, _SRC, _f32, _DST, _i32, _i32, _i32, _P]),
    "dvc_mul_act_grad_add": (C.c_int, [_SRC, _SRC, _f32, _SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_zero_upsample2": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _i32, _i32, _P]),
    "dvc_add_scaled": (C.c_int, [_SRC, _SRC, _P, _SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_scale_planes": (C.c_int, [_SRC, _P, _DST, _i32, _i32, _i32, _P]),
    "dvc_plane_dot_scratch_floats": (_i64, [_i32, _i32]),
    "dvc_plane_dot": (C.c_int, [_SRC, _SRC, _i32, _i32, _i32, _i32, _P, _P, _P]),
    "dvc_flow_warp_backward": (C.c_int, [_SRC, _SRC, _SRC, _DST, _DST, _i32, _i32, _i32, _P]),
    "dvc_avg_pool2_backward": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _f32, _P]),
    "dvc_max_pool2_backward": (C.c_int, [_SRC, _SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_bilinear_up2_backward": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _f32, _P]),
    "dvc_eb_backward_scratch_floats": (_i64, [_i32]),
    "dvc_eb_backward": (C.c_int, [_SRC, _P, _SRC, _DST, _P, _P, _i32, _i32, _i32, _P]),
    "dvc_dual_prior_pass1_backward": (C.c_int, [_SRC, _DST, _i32, _i32, _i32, _P]),
    "dvc_dual_prior_pass2_backward": (C.c_int, [_SRC, _SRC, _SRC, _SRC, _P, _SRC, _SRC, _DST, _DST, _DST, _DST,
                                                _i32, _i32, _i32, _P]),
    "dvc_likelihood_bits_backward": (C.c_int, [_SRC, _P, _DST, _i32, _i32, _i32, _P]),
    "dvc_sq_err_backward": (C.c_int, [_SRC, _SRC, _P, _DST, _i32, _i32, _i32, _P]),
    "dvc_rans_encode": (_i64, [_P, _P, _i64, _P, _i32, _i32, _P, _P, _P, _i64]),
    "dvc_rans_decode": (C.c_int, [_P, _i64, _P, _i64, _P, _i32, _i32, _P, _P, _P]),
    "dvc_pmf_to_quantized_cdf": (C.c_int, [_P, _i32, _i32, _P]),
    "dvc_last_error": (C.c_char_p, []),
    "dvc_device_count": (C.c_int, []),
    "dvc_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libdvc_hip.so once; raises DvcError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DvcError(
            f"{LIB_PATH} is missing: build it with `make -C deepvideocodec_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().dvc_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc < 0:
        raise DvcError(f"{what} failed (code {rc}): {last_error()}")
    return rc
