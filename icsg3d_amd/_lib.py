"""ctypes binding of libicsg3d_hip.so (include/icsg3d.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C icsg3d_amd/csrc``.  There
is NO CPU fallback: if the shared object is missing or fails to load, importing a compute entry
point raises ``IcsLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICSG3D_LIB_PATH: another build of the same library (kernel A/B experiments: scripts/variants.sh)
LIB_PATH = os.environ.get("ICSG3D_LIB_PATH") or os.path.join(_HERE, "libicsg3d_hip.so")


class IcsLibraryError(RuntimeError):
    pass


class IcsError(RuntimeError):
    pass


class UnetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("num_classes", C.c_int), ("d", C.c_int),
                ("max_batch", C.c_int), ("lr", C.c_float), ("loss_weight", C.c_float),
                ("pool_ties_all", C.c_int), ("bn_unbias", C.c_int), ("bce_from_logits", C.c_int)]


class VaeConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("cond_shape", C.c_int), ("latent_dim", C.c_int),
                ("filters", C.c_int * 4), ("d", C.c_int), ("max_batch", C.c_int), ("lr", C.c_float),
                ("alpha", C.c_float), ("beta", C.c_float), ("pm_layer_weights", C.c_float * 4),
                ("bn_unbias", C.c_int)]


_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)
_I64 = C.POINTER(C.c_int64)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol include/icsg3d.h declares
SIGNATURES = {
    "ics_last_error": (C.c_char_p, []),
    "ics_version": (C.c_char_p, []),
    "ics_kernel_launches": (C.c_longlong, []),
    "ics_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ics_set_device": (C.c_int, [C.c_int]),
    "ics_device_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "ics_unet_create": (C.c_int, [C.POINTER(UnetConfig), C.POINTER(_H)]),
    "ics_unet_predict": (C.c_int, [_H, _F, C.c_int, _F, _F]),
    "ics_unet_predict_labels": (C.c_int, [_H, _F, C.c_int, C.c_float, _U8, _U8]),
    "ics_unet_train_step": (C.c_int, [_H, _F, _U8, C.c_int, _F]),
    "ics_unet_test_step": (C.c_int, [_H, _F, _U8, C.c_int, _F]),
    "ics_unet_upload_batch": (C.c_int, [_H, _F, _U8, C.c_int]),
    "ics_unet_train_step_resident": (C.c_int, [_H, _F]),
    "ics_vae_create": (C.c_int, [C.POINTER(VaeConfig), _H, C.POINTER(_H)]),
    "ics_vae_encode": (C.c_int, [_H, _F, _F, _F, C.c_int, _F, _F, _F]),
    "ics_unet_predict_resident": (C.c_int, [_H, C.c_int, C.c_float]),
    "ics_net_share_stream": (C.c_int, [_H, _H]),
    "ics_net_timer_start": (C.c_int, [_H]),
    "ics_net_timer_stop": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "ics_unet_metric_sums": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "ics_vae_decode": (C.c_int, [_H, _F, _F, C.c_int, _F]),
    "ics_vae_train_step": (C.c_int, [_H, _F, _F, _F, C.c_int, _F]),
    "ics_vae_test_step": (C.c_int, [_H, _F, _F, _F, C.c_int, _F]),
    "ics_vae_upload_batch": (C.c_int, [_H, _F, _F, _F, C.c_int]),
    "ics_vae_train_step_resident": (C.c_int, [_H, _F]),
    "ics_vae_decode_to_unet_labels": (C.c_int, [_H, _H, _F, _F, C.c_int, C.c_float, _U8, _U8, _F, _F]),
    "ics_vae_decode_to_unet_atoms": (C.c_int, [_H, _H, _F, _F, C.c_int, C.c_float, C.c_int, C.c_int, _U8, _U8, _F, _F,
                                               _I32, _I32, _I32, _I64]),
    "ics_op_segment_atoms": (C.c_int, [_U8, _U8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _I32, _I32, _I32, _I64]),
    "ics_release_caches": (C.c_int, []),
    "ics_op_label_boxes": (C.c_int, [_I32, _I32, C.c_int, C.c_int, C.c_int, _I32, _I32, _I32]),
    "ics_op_watershed_split": (C.c_int, [_I32, _I32, _I32, C.c_int, C.c_int, _I32]),
    "ics_op_component_bounds": (C.c_int, [_I32, _I32, C.c_int, _I32, _I32, C.c_int, C.c_int, C.c_double, _I64]),
    "ics_op_region_stats": (C.c_int, [_I32, _U8, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _I32]),
    "ics_net_destroy": (C.c_int, [_H]),
    "ics_net_sync": (C.c_int, [_H]),
    "ics_net_num_tensors": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "ics_net_tensor_info": (C.c_int, [_H, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "ics_net_set_tensor": (C.c_int, [_H, C.c_char_p, _F, C.c_size_t]),
    "ics_net_get_tensor": (C.c_int, [_H, C.c_char_p, _F, C.c_size_t]),
    "ics_net_get_grad": (C.c_int, [_H, C.c_char_p, _F, C.c_size_t]),
    "ics_net_get_activation": (C.c_int, [_H, C.c_char_p, _F, C.c_size_t]),
    "ics_net_get_bn_affine": (C.c_int, [_H, C.c_char_p, _F, _F, C.c_size_t]),
    "ics_net_check_canaries": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "ics_net_set_lr": (C.c_int, [_H, C.c_float]),
    "ics_net_reset_optimizer": (C.c_int, [_H]),
    "ics_net_num_params": (C.c_int, [_H, C.POINTER(C.c_size_t)]),
    "ics_net_get_optimizer_state": (C.c_int, [_H, _F, _F, C.c_size_t, C.POINTER(C.c_int)]),
    "ics_net_set_optimizer_state": (C.c_int, [_H, _F, _F, C.c_size_t, C.c_int]),
    "ics_net_graph_probe": (C.c_int, [_H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "ics_net_profile_enable": (C.c_int, [_H, C.c_int]),
    "ics_net_profile_filter": (C.c_int, [_H, C.c_char_p]),
    "ics_net_profile_count": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "ics_net_profile_row": (C.c_int, [_H, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ics_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ics_net_comm_init": (C.c_int, [_H, C.c_int, C.c_int, C.c_char_p]),
    "ics_net_comm_allreduce_max": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "ics_net_comm_broadcast_state": (C.c_int, [_H, C.c_int]),
    "ics_net_set_sync_bn": (C.c_int, [_H, C.c_int]),
    "ics_net_comm_info": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ics_op_unet_head": (C.c_int, [_F, _F, _F, _F, _F, _U8, C.c_size_t, C.c_int, C.c_float, C.c_int, C.c_int, _F, _F,
                                  C.POINTER(C.c_double)]),
    "ics_op_conv3d_forward": (C.c_int, [_F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]),
    "ics_op_conv3d_bench": (C.c_int, [C.c_int] * 8 + [_F]),
    "ics_op_conv3d_backward": (C.c_int, [_F, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F, _F]),
}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes.  Raises IcsLibraryError loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IcsLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C icsg3d_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    # The engines run up to five HIP streams at once in one process (U-Net: compute + RCCL; DFC-VAE: compute + second stream +
    # RCCL).  The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two of OUR streams on
    # one queue serialise and the two-stream schedule of the DFC-VAE step loses 10 % instead of gaining 7 % (measured,
    # DESIGN.md section 10).  Read by the runtime when HIP initialises, i.e. at the first HIP call -- which this library makes.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    except OSError as e:  # pragma: no cover
        raise IcsLibraryError("failed to load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise IcsLibraryError("libicsg3d_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ics_last_error()
        raise IcsError(msg.decode() if msg else "icsg3d call failed (%d)" % rc)


def fptr(a):
    return a.ctypes.data_as(_F) if a is not None else None


def u8ptr(a):
    return a.ctypes.data_as(_U8) if a is not None else None


def i32ptr(a):
    return a.ctypes.data_as(_I32) if a is not None else None


def i64ptr(a):
    return a.ctypes.data_as(_I64) if a is not None else None


def device_count():
    n = C.c_int(0)
    lib = load()
    if lib.ics_device_count(C.byref(n)) != 0:
        return 0
    return n.value
