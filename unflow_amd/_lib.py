"""ctypes binding of libunflow_hip.so (the C ABI declared in include/unflow_hip.h).

Mirrors what src/e2eflow/ops.py:56-63 does with tf.load_op_library, minus the
compile-on-import: the library is built ahead of time by ``make -C
unflow_amd/csrc`` (``__graft_entry__.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNFLOW_HIP_LIB: another build of the SAME library (tools/variants.sh builds A/B variants of a
# kernel with -D knobs); never a fallback -- a missing file raises like the default path does.
LIB_PATH = os.environ.get("UNFLOW_HIP_LIB") or os.path.join(_HERE, "libunflow_hip.so")

ABI_VERSION = 5      # UNFLOW_ABI_VERSION of include/unflow_hip.h
UNFLOW_OK = 0
UNFLOW_ERR_INVALID_ARGUMENT = 1
UNFLOW_ERR_LAUNCH = 2

_I = ctypes.c_int
_P = ctypes.c_void_p
_F = ctypes.c_float
_LL = ctypes.c_longlong

# symbol -> (restype, argtypes); the single source of truth for the ABI on the
# Python side (tests check it against include/unflow_hip.h).
SIGNATURES = {
    "unflow_abi_version": (_I, []),
    "unflow_last_error": (ctypes.c_char_p, []),
    "unflow_build_flags": (ctypes.c_char_p, []),
    "unflow_stream_copy": (_I, [_P, _P, ctypes.c_size_t, _P]),
    "unflow_correlation_geometry": (_I, [_I, _I, _I, _I, _I, _I, _I, ctypes.POINTER(_I)]),
    "unflow_correlation_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_correlation_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_backward_warp_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_backward_warp_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_image_warp_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_image_warp_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_forward_warp_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "unflow_forward_warp_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "unflow_downsample_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "unflow_downsample_pyramid": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "unflow_reduce_workspace_floats": (ctypes.c_size_t, [_I, _I, _I]),
    "unflow_reduce_plan_begin": (_I, [_P]),
    "unflow_reduce_plan_end": (_I, []),
    "unflow_reduce_plan_run": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "unflow_sum_tensors": (_I, [_P, _I, _P, _LL, _P]),
    "unflow_scale_stack_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "unflow_scale_stack_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "unflow_ternary_loss_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _LL, _P]),
    "unflow_ternary_loss_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_ternary_loss_warp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _LL, _P]),
    "unflow_ternary_loss_warp_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_outgoing_mask": (_I, [_P, _P, _I, _I, _I, _P]),
    "unflow_create_mask": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_fb_occlusion": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "unflow_fb_occlusion_masked": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "unflow_fb_occlusion_losses": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _P, _P, _LL, _I, _I, _I, _P]),
    "unflow_fb_occlusion_warp_losses": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _LL, _P, _P, _LL, _I, _I, _I, _P]),
    "unflow_fb_occlusion_warp": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_sym_loss_fwd": (_I, [_P, _P, _F, _P, _P, _I, _I, _I, _LL, _P]),
    "unflow_charbonnier_fwd": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _LL, _P]),
    "unflow_charbonnier_bwd": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _LL, _P]),
    "unflow_smoothness_loss_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _LL, _P]),
    "unflow_smoothness_loss_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _LL, _P]),
    "unflow_smoothness_deltas": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_gradient_loss_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _LL, _P]),
    "unflow_gradient_loss_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _LL, _P]),
    "unflow_gradient_delta": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "unflow_correlation_flip": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_correlation_fwd_f16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_correlation_bwd_f16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_correlation_fwd_f16_exact": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_correlation_bwd_f16_exact": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "unflow_backward_warp_fwd_f16": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_backward_warp_bwd_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_image_warp_fwd_f16": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_image_warp_bwd_flow_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "unflow_forward_warp_fwd_f16": (_I, [_P, _P, _I, _I, _I, _P]),
    "unflow_census_transform": (_I, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "unflow_census_hamming": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "unflow_stacked_inputs_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "unflow_resize_bilinear_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "unflow_resize_bilinear_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "unflow_resize_area_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
}

PLAN_CAPACITY = 128   # UNFLOW_PLAN_CAPACITY
PLAN_TERMS = 8        # UNFLOW_PLAN_TERMS: floats of the per-term buffer of unflow_reduce_plan_run


class ReducePlan(ctypes.Structure):
    """unflow_reduce_plan of include/unflow_hip.h: caller-owned list of deferred loss reductions."""
    _fields_ = [("n", _I), ("partial", _P * PLAN_CAPACITY), ("count", _I * PLAN_CAPACITY),
                ("inv_norm", ctypes.c_double * PLAN_CAPACITY)]


_lib = None


class UnflowLibraryError(ImportError):
    pass


def load():
    """Load libunflow_hip.so; raise (never fall back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UnflowLibraryError(
                "libunflow_hip.so not found at %s -- build it with `make -C unflow_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        # torch first: it ships its own libamdhip64.so.7, and the HIP runtime that is loaded
        # first serves the whole process -- the tensors (torch) and these kernels must share it
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.unflow_abi_version() != ABI_VERSION:
            raise UnflowLibraryError("libunflow_hip.so ABI version mismatch")
        flags = lib.unflow_build_flags().decode().strip()
        if flags and os.environ.get("UNFLOW_ALLOW_DIAGNOSTIC_BUILD") != "1":
            # a timing-only variant (tools/variants.sh), some of which compute wrong results on purpose
            raise UnflowLibraryError("%s is a diagnostic build (%s): set UNFLOW_ALLOW_DIAGNOSTIC_BUILD=1 to load it"
                                     % (LIB_PATH, flags))
        _lib = lib
    return _lib


def last_error():
    return load().unflow_last_error().decode()


def check(rc):
    """Map a C status to the exception the reference's OP_REQUIRES would raise
    (tf.errors.InvalidArgumentError -> ValueError here)."""
    if rc == UNFLOW_OK:
        return
    msg = last_error()
    if rc == UNFLOW_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    raise RuntimeError("unflow_hip: " + msg)
