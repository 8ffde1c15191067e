"""Prototypes of the scorer / fused-step entry points (include/ltr_mi355x.h, second half)."""
from ctypes import c_float, c_int, c_int64, c_uint64, c_void_p

P = c_void_p
PROTOTYPES = {
    "ltr_net_info": (c_int, [c_int, P]),
    "ltr_fused_grid": (c_int, [c_int, c_int]),
    "ltr_mlp_pack": (c_int, [c_int, P, P, P, P, P, P, P, P]),
    "ltr_mlp_pack_sub": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P]),
    "ltr_mlp_reduce_grads_sub": (c_int, [c_int, c_int, c_int, c_int, P, c_int, P, P]),
    "ltr_dropout_keep_mask": (c_int, [c_uint64, c_int, c_int64, c_int, P, P]),
    "ltr_dropout_keep_mask_p": (c_int, [c_uint64, c_int, c_int64, c_int, c_float, P, P]),
    "ltr_mlp_forward": (c_int, [c_int, P, c_int64, P, c_int, c_uint64, P, P, P, c_int, P]),
    "ltr_mlp_backward": (c_int, [c_int, P, c_int64, P, c_int, c_uint64, P, P, P, P, c_int, P]),
    "ltr_mlp_reduce_grads": (c_int, [c_int, P, c_int, P, P]),
    "ltr_mlp_acts_floats": (c_int64, [c_int, c_int64]),
    "ltr_mlp_forward_save": (c_int, [c_int, P, c_int64, P, c_int, c_uint64, P, P, P, P, c_int, P]),
    "ltr_mlp_backward_saved": (c_int, [c_int, P, c_int64, P, c_int, P, P, P, c_int, P]),
    "ltr_fused_step": (c_int, [c_int, c_int, P, P, c_int, c_int, P, c_int, c_uint64, P, P, c_float, c_float, c_float,
                               c_int, c_float, P, P, c_int, P]),
    "ltr_triple_fold": (c_int, [P, P, P, P, P, c_int, c_int, P, P, P, P]),
    "ltr_triple_unfold_grads": (c_int, [P, c_int, c_int, P, P, P, P, P]),
    "ltr_debug_set_stamps": (c_int, [P, c_int]),
    "ltr_fused_step_lambda": (c_int, [c_int, P, P, c_int, c_int, P, c_int, c_uint64, P, P, c_int, c_int, c_float, c_float,
                                      c_float, c_float, c_int, c_float, P, P, P, c_int, P]),
}


def bind(h):
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            from ._lib import LtrBuildError
            raise LtrBuildError(f"library does not export {name}: stale build?") from e
        fn.restype = res
        fn.argtypes = args
