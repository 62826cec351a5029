"""ctypes binding of libdrs_hip.so (include/drs.h).  There is no CPU fallback: if the HIP
library is missing or a symbol is absent this module raises, and so does everything above it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdrs_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "libdrs_hip_dev.so")     # the same sources with -DDRS_DEV: + the switches of include/drs_dev.h

_p, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_sz, _u64 = C.c_size_t, C.c_ulonglong

# name -> (restype, argtypes); mirrors include/drs.h declaration by declaration
SIGNATURES = {
    "drs_conv_mtile": (_i, [_i]),
    "drs_conv_forward": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p]),
    "drs_conv_workspace_floats": (_sz, [_i]),
    "drs_conv_halo_skip": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "drs_conv_forward_ws": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p, _sz, _p]),
    "drs_conv_wgrad_splits": (_i, [_i, _i, _i, _i, _i]),
    "drs_conv_wgrad": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "drs_filter_flip_transpose": (_i, [_p, _p, _i, _i, _i, _p]),
    "drs_filter_pad_cin": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "drs_split_conv_mtile": (_i, [_i]),
    "drs_split_terms": (_i, [_p, _sz, _i, _p, _p]),
    "drs_filter_split": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "drs_conv_forward_split": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _i, _p]),
    "drs_conv_wgrad_split_splits": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "drs_conv_wgrad_split": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "drs_colsum_scratch_doubles": (_i, [_i]),
    "drs_stats_reduce": (_i, [_p, _i, _i, _p, _p, _p]),
    "drs_stats_reduce_means": (_i, [_p, _i, _i, _d, _p, _p, _p]),
    "drs_conv_stats_reduce": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "drs_conv_stats_finish": (_i, [_p, _i, _i, _i, _d, _p, _p, _p, _d, _i, _p, _p]),
    "drs_bn_finish": (_i, [_p, _d, _i, _p, _p, _p, _d, _i, _p]),
    "drs_bn_eval_coeffs": (_i, [_p, _p, _i, _p, _p]),
    "drs_bn_act_pool_forward": (_i, [_p, _i, _i, _i, _p, _f, _i, _p, _i, _i, _i, _p, _p]),
    "drs_bn_finish_act_pool_forward": (_i, [_p, _d, _p, _p, _p, _d, _i, _p, _i, _i, _i, _f, _i, _p, _i, _i, _i, _p, _p]),
    "drs_bn_act_pool_forward_terms": (_i, [_p, _i, _i, _i, _p, _f, _i, _p, _i, _i, _i, _p, _p, _i, _p]),
    "drs_bn_backward_rows": (_i, [_i, _i, _i, _i]),
    "drs_bn_backward_reduce": (_i, [_p, _i, _i, _p, _p, _i, _i, _i, _p, _f, _i, _p, _p, _p]),
    "drs_bn_backward_apply": (_i, [_p, _p, _i, _i, _i, _p, _p, _d, _p, _i, _i, _i, _p]),
    "drs_bn_backward_apply_means": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _p]),
    "drs_bn_backward_apply_terms": (_i, [_p, _p, _i, _i, _i, _p, _p, _d, _p, _i, _i, _i, _p, _i, _p]),
    "drs_avg_pool_forward": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _i, _p]),
    "drs_avg_pool_backward": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "drs_se_forward": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "drs_se_backward": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "drs_classifier_rows": (_i, [_i, _i]),
    "drs_classifier_loss": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _i, _i, _p, _p, _p,
                                 _p, _p]),
    "drs_rows_reduce_f32": (_i, [_p, _i, _i, _p, _p, _p]),
    "drs_sum_f64": (_i, [_p, _i, _p, _p]),
    "drs_l2_loss": (_i, [_p, _sz, _p, _p, _p]),
    "drs_momentum_update": (_i, [_p, _p, _p, _sz, _sz, _f, _f, _f, _f, _p]),
    "drs_confusion": (_i, [_p, _p, _p, _sz, _i, _i, _p, _p]),
    "drs_crop_normalize": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _u64, _i, _p, _p, _i, _i, _i, _i, _p, _p,
                                _p, _i, _i, _p]),
    "drs_stitch_accumulate": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "drs_stitch_finalize": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "drs_softmax_accumulate": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "drs_scale_f64": (_i, [_p, _i, _d, _p]),
    # ---- step level (csrc/engine.hip)
    "drs_net_create": (_i, [C.c_char_p, _i, _i, _f, _i, _i, _i, _f, C.POINTER(_p)]),
    "drs_net_destroy": (None, [_p]),
    "drs_net_num_buffers": (_i, [_p]),
    "drs_net_buffer_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_sz), C.POINTER(_i)]),
    "drs_net_bind": (_i, [_p, C.c_char_p, _p, _sz]),
    "drs_net_buffer": (_i, [_p, C.c_char_p, C.POINTER(_p), C.POINTER(_sz)]),
    "drs_net_layout": (_i, [_p, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "drs_net_layer_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_i), C.c_char_p, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i)]),
    "drs_net_info": (_i, [_p, C.c_char_p, _i, C.POINTER(_f), C.POINTER(_i), C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i)]),
    "drs_net_se_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "drs_net_type_name": (_i, [_i, C.c_char_p, _i, C.POINTER(_i)]),
    "drs_net_num_variables": (_i, [_p]),
    "drs_net_variable_info": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i), C.POINTER(_i)]),
    "drs_params_get": (_i, [_p, C.c_char_p, C.c_char_p, _p, _sz, _p]),
    "drs_params_set": (_i, [_p, C.c_char_p, C.c_char_p, _p, _sz, _p]),
    "drs_grad_buffer": (_i, [_p, C.POINTER(_p), C.POINTER(_sz)]),
    "drs_net_global_step": (C.c_longlong, [_p, C.c_longlong]),
    "drs_net_learning_rate": (_f, [_p, _f]),
    "drs_net_set_comm": (_i, [_p, _i, _i, _p, _p, _p]),
    "drs_rccl_bind_library": (_i, [C.c_char_p]),
    "drs_rccl_available": (_i, []),
    "drs_rccl_form": (_i, []),
    "drs_rccl_unique_id": (_i, [_p]),
    "drs_rccl_comm_create": (_i, [_i, _i, _p, C.POINTER(_p)]),
    "drs_rccl_comm_destroy": (_i, [_p]),
    "drs_rccl_all_reduce": (_i, [_p, _p, C.c_size_t, _i, _p]),
    "drs_net_set_rccl": (_i, [_p, _i, _i, _p, _p, _p]),
    "drs_train_step": (_i, [_p, _i, _i, _f, _i, _d, _p]),
    "drs_forward": (_i, [_p, _i, _i, _i, _i, _p]),
    "drs_apply_update": (_i, [_p, _f, _p]),
    "drs_net_set_two_streams": (_i, [_p, _i]),
    "drs_net_timing": (_i, [_p, _i]),
    "drs_net_num_timing_kinds": (_i, []),
    "drs_net_timing_summary": (_i, [_p, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_d), C.POINTER(_d)]),
}
# callback types of drs_net_set_comm
ALLREDUCE_FN = C.CFUNCTYPE(_i, _p, _p, _sz, _i, _i, _p)
WAIT_FN = C.CFUNCTYPE(_i, _p, _i, _p)
WANT_LOGITS, WITH_LABELS, USE_ACC_MASK, USE_LOSS_MASK, NO_UPDATE = 1, 2, 4, 8, 16


# include/drs_dev.h (libdrs_hip_dev.so only)
DEV_SIGNATURES = {
    "drs_debug_skip_taps": (_i, [_i]), "drs_debug_conv_variant": (_i, [_i]), "drs_debug_conv_wide192": (_i, [_i]),
    "drs_debug_conv_splitk": (_i, [_i]), "drs_debug_conv_hybrid": (_i, [_i]), "drs_debug_conv_sk_order": (_i, [_i]), "drs_debug_conv_prio": (_i, [_i]),
    "drs_debug_conv_sk_geometry": (_i, [_i, _i, _i, _p]), "drs_debug_conv_trace": (_i, [_p]), "drs_debug_conv_lpt": (_i, [_i]), "drs_debug_conv_order": (_i, [_i] * 7 + [_p, _i]),
    "drs_debug_wgrad_variant": (_i, [_i]), "drs_debug_wgrad_seg": (_i, [_i]), "drs_debug_wgrad_balance": (_i, [_i]), "drs_debug_wgrad_target": (_i, [_i]),
    "drs_debug_wgrad_target_big": (_i, [_i]), "drs_debug_wgrad_len": (_i, [_i]), "drs_debug_wgrad_minchunks": (_i, [_i]), "drs_debug_wgrad_model": (_i, [_i]), "drs_debug_wgrad_ablate": (_i, [_i]), "drs_debug_wgrad_prio": (_i, [_i]), "drs_debug_cls_variant": (_i, [_i]), "drs_debug_slide_blocks": (_i, [_i]), "drs_debug_slide_minrows": (_i, [_i]), "drs_debug_slide_rowpad": (_i, [_i]), "drs_debug_chain_mode": (_i, [_i]), "drs_debug_jitter": (_i, [C.c_ulonglong]), "drs_debug_wg_stream_prio": (_i, [_i]), "drs_debug_reductions_on_chain": (_i, [_i]), "drs_debug_wgrad_schedule": (_i, [_i, _i, _i]), "drs_debug_variant": (_i, [_i]),
    "drs_debug_wgrad_cut": (_i, [_i] * 7 + [_p, _i, _p, _p]),
}


class DrsError(RuntimeError):
    pass


_lib = None
_dev = None


def _open(path, signatures):
    if not os.path.isfile(path):
        raise DrsError("HIP library not built: %s is missing (run dynamic-rs-segmentation_amd/csrc/build.sh "
                       "or __graft_entry__.build()); there is no CPU fallback" % path)
    # PyTorch supplies device memory and streams, so the library must share ITS HIP runtime: import torch first
    # (libdrs_hip.so then binds to the already-loaded libamdhip64.so.7 instead of pulling in a second copy)
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """dlopen the library once and type every symbol of include/drs.h."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH, SIGNATURES)
    return _lib


class _Dev(object):
    """the development build of the library (include/drs_dev.h): same entry points + the A/B switches.  Used by tools/ and by the
    tests that hold two kernel forms equal; the package's product path never touches it."""

    def __init__(self, lib):
        self.lib = lib

    def __getattr__(self, name):
        return getattr(self.lib, name)

    def load(self):
        return self.lib

    def call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc != 0:
            raise DrsError("%s -> %s" % (name, _STATUS.get(rc, rc)))

    def query(self, name, *args):
        return getattr(self.lib, name)(*args)


def dev():
    global _dev
    if _dev is None:
        sig = dict(SIGNATURES)
        sig.update(DEV_SIGNATURES)
        _dev = _Dev(_open(DEV_LIB_PATH, sig))
    return _dev


_STATUS = {1: "DRS_ERR_ARG (rejected argument)", 2: "DRS_ERR_HIP (launch failed)"}


def call(name, *args):
    """Call an int-status entry point; raise on a non-zero status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise DrsError("%s -> %s" % (name, _STATUS.get(rc, rc)))


def query(name, *args):
    """Call a pure size query (returns its int)."""
    return getattr(load(), name)(*args)
