"""ctypes binding of libgnm_hip.so (include/gnm_hip.h).

There is deliberately no fallback: if the library is missing or a symbol does not
resolve, importing this module raises, and every call checks the returned status.
"""
import ctypes as C
import os

from . import _build

_c_f32p = C.c_void_p   # device pointers travel as plain integers (tensor.data_ptr())
_i, _ll, _f, _p = C.c_int, C.c_longlong, C.c_float, C.c_void_p

# name -> (restype, [argtypes]); order and meaning exactly as in include/gnm_hip.h
SIGNATURES = {
    "gnm_version": (C.c_char_p, []),
    "gnm_debug_device_once": (_i, [_i, _i]),
    "gnm_csr_from_edge_mat": (_i, [_p, _ll, _i, _p, _p]),
    "gnm_csr_transpose": (_i, [_p, _p, _i, _p, _p]),
    "gnm_csr_parity_order": (_i, [_p, _p, _i]),
    "gnm_csr_is_symmetric": (_i, [_p, _p, _i]),
    "gnm_batch_coo_from_csr": (_ll, [_p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "gnm_agg": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _i, _p, _i, _i, _i, _p, _i, _p, _p]),
    "gnm_agg_bwd_stats": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _i, _p, _i, _i, _p, _i, _p,
                               _p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p]),
    "gnm_agg_fwd_bnrelu": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _i, _p, _i, _i, _p, _i, _i, _p, _i,
                                _i, _p]),
    "gnm_agg_slice_width": (_i, [_i, _i]),
    "gnm_agg_num_partials": (_i, [_i, _i, _i]),
    "gnm_sum_partials": (_i, [_p, _i, _p, _p]),
    "gnm_adj_bits_words": (_ll, [_i]),
    "gnm_aggm_max_nodes": (_i, []),
    "gnm_aggm_num_partials": (_i, [_i, _i]),
    "gnm_adj_bits_build": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "gnm_aggm": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _i, _i, _i, _p, _i, _p, _p]),
    "gnm_aggm_bwd_stats": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _p, _i, _i, _p, _i, _p,
                                _p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p]),
    "gnm_aggm_fwd_bnrelu": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p, _i, _p, _i, _i, _p, _i, _i, _p,
                                 _i, _i, _p]),
    "gnm_rowdot_num_partials": (_i, []),
    "gnm_rowdot_partials": (_i, [_p, _i, _p, _i, _ll, _i, _p, _p]),
    "gnm_sum_partials_multi": (_i, [_p, _ll, _p, _i, _p, _p]),
    "gnm_maxpool_colmin_blocks": (_i, [_i]),
    "gnm_maxpool_colmin": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "gnm_maxpool_fwd": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "gnm_maxpool_bwd": (_i, [_p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _p, _p, _i, _p]),
    "gnm_maxpool_fwd_tiled": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "gnm_maxpool_bwd_tiled": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p]),
    "gnm_linear_grid": (_i, [_i]),
    "gnm_linear_max_k": (_i, [_i]),
    "gnm_linear_fwd": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p, _p]),
    "gnm_wgrad_grid": (_i, [_i]),
    "gnm_wgrad_workspace_floats": (_ll, [_i, _i, _i]),
    "gnm_linear_wgrad": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p, _p]),
    "gnm_linear_bwd_grid": (_i, [_i]),
    "gnm_linear_bwd_workspace_floats": (_ll, [_i, _i, _i]),
    "gnm_linear_bwd_fused": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p,
                                  _p, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "gnm_linear_bwd_fused_rz": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _p,
                                     _p, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "gnm_linear_dgrad_masked": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "gnm_debug_lin_first_tile": (_i, [_i, _i, _i, _i, _i]),
    "gnm_small_gemm": (_i, [_p, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "gnm_reduce_partials_multi": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "gnm_bn_finalize": (_i, [_p, _i, _i, _ll, _p, _p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _p]),
    "gnm_gather_graph_rows": (_i, [_p, _p, _i, _i, _p, _p, _i, _p, _p, _i, _p]),
    "gnm_bn_relu_readout": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _p, _i, _i, _p]),
    "gnm_bn_relu_bwd_stats": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p,
                                   _i, _i, _p, _p]),
    "gnm_bn_bwd_finalize": (_i, [_p, _i, _i, _ll, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    "gnm_bn_bwd_apply": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _ll, _i, _p]),
    "gnm_eval_max_nodes": (_i, []),
    "gnm_eval_table_words": (_ll, [_i, _i]),
    "gnm_eval_encoder": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _ll, _i, _p,
                              _p, _i, _p, _i, _p, _p, _i, _p]),
    "gnm_eval_layers_scratch_floats": (_ll, [_i, _i, _i, _i]),
    "gnm_eval_layers": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _ll, _i, _p,
                             _p, _i, _p, _p, _i, _p]),
    "gnm_disc_score_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _i, _i, _p, _p]),
    "gnm_disc_score_fwd_unit": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _p, _p]),
    "gnm_disc_unit_scale": (_i, [_p, _i, _i, _p, _f, _i, _p, _i, _p, _p, _p, _p]),
    "gnm_disc_score_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p, _p]),
    "gnm_head_fwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _i, _p]),
    "gnm_head_bwd": (_i, [_p, _i, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    "gnm_loss_workspace_doubles": (_ll, [_ll]),
    "gnm_loss_ce_bce": (_i, [_p, _i, _p, _i, _i, _p, _p, _ll, _ll, _f, _p, _p, _i, _p, _p, _p]),
    "gnm_loss_ce_bce_grad": (_i, [_p, _i, _p, _i, _i, _p, _p, _ll, _ll, _f, _p, _p, _i, _p, _p]),
    "gnm_adam_step": (_i, [_p, _p, _p, _p, _ll, _p, _p, _p]),
}


class GnmError(RuntimeError):
    pass


def _load():
    path = _build.LIB_PATH
    if not os.path.exists(path):
        raise GnmError(
            "libgnm_hip.so not found at %s -- build it first (python __graft_entry__.py, or "
            "python graph-neural-mapping_amd/gnm/_build.py).  There is no CPU fallback." % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME
    # as /opt/rocm's).  Load torch's copy FIRST so the dynamic loader resolves this
    # library's DT_NEEDED libamdhip64.so.7 to it; otherwise streams and device pointers
    # would cross two independent runtimes (hipErrorNoDevice / invalid handle).
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(path)
    variant = bool(os.environ.get("GNM_HIP_LIB"))    # an A/B build of an older commit may predate a symbol
    for name, (res, args) in SIGNATURES.items():
        if variant and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

_STATUS = {-1: "GNM_ERR_BAD_ARG", -2: "GNM_ERR_UNSUPPORTED"}


def check(status, what):
    if status != 0:
        raise GnmError("%s failed: %s" % (what, _STATUS.get(status, "hipError %d" % status)))


def ptr(t):
    """device/host pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()
