"""Prototypes of the scorer / fused-step entry points (include/ltr_mi355x.h, second half)."""
from ctypes import c_float, c_int, c_int64, c_uint64, c_void_p

P = c_void_p
PROTOTYPES = {}


def bind(h):
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            from ._lib import LtrBuildError
            raise LtrBuildError(f"library does not export {name}: stale build?") from e
        fn.restype = res
        fn.argtypes = args
