"""ctypes binding of libltr_mi355x.so (the C ABI declared in include/ltr_mi355x.h)."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTR_LIB: alternative build of the same ABI (kernel A/B experiments); default is the in-tree library.
_SO = os.environ.get("LTR_LIB") or os.path.join(_HERE, "libltr_mi355x.so")


class LtrError(RuntimeError):
    """A launcher returned non-zero (argument rejected or hipError_t)."""


class LtrBuildError(LtrError):
    """libltr_mi355x.so is missing / not loadable: the HIP path is the only path, so this is fatal."""


class LtrDeviceError(LtrError):
    """An op was handed a tensor that does not live on a ROCm device."""


P = c_void_p
_PROTOTYPES = {
    "ltr_abi_version": (c_int, []),
    "ltr_error_string": (c_char_p, [c_int]),
    "ltr_reduce_sum_f32": (c_int, [P, c_int64, c_float, P, P]),
    "ltr_approxndcg_fwd_bwd": (c_int, [P, P, c_int, c_int, c_float, c_float, c_float, c_float, P, P, P]),
    "ltr_listnet_fwd_bwd": (c_int, [P, P, c_int, c_int, c_int, c_float, P, P, P]),
    "ltr_lambda_fwd_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int,
                                   c_float, P, P, P, P]),
    "ltr_lambda_pairs_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int,
                                     P, P, P, P]),
    "ltr_lambda_pairs_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int,
                                     P, P, P]),
    "ltr_lambda_colsum_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P, P]),
    "ltr_lambda_colsum_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int,
                                      P, P, P]),
    "ltr_lambda_colsum_sys_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P, P]),
    "ltr_lambda_colsum_sys_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P, P, P]),
    "ltr_risk_fwd_bwd": (c_int, [P, c_int, c_int, c_int, c_float, c_int, P, P, P]),
    "ltr_trisk_fwd_bwd": (c_int, [P, P, c_int, c_float, P, P, P, P]),
    "ltr_trisk_tail_fwd_bwd": (c_int, [P, c_int, c_float, c_int, c_float, P, P, P]),
    "ltr_risk_tail_fwd_bwd": (c_int, [P, c_int, c_int, c_float, c_int, c_int, c_int, c_float, c_int, P, P, P]),
    "ltr_risk_matrix_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "ltr_ndcg_at_k": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "ltr_svmlight_scan": (c_int, [c_char_p, P, P, P, c_int]),
    "ltr_svmlight_load": (c_int, [c_char_p, c_int64, c_int, c_int, P, P, P, c_int]),
    "ltr_gather_rows_f32": (c_int, [P, c_int64, P, c_int64, c_int64, P, P]),
    # include/ltr_encoder.h (row f-3)
    "ltr_enc_cast_bf16": (c_int, [P, P, c_int64, P]),
    "ltr_enc_splitk_epilogue": (c_int, [P, c_int, c_int64, c_int, P, c_float, c_uint64, c_int, P, P, P]),
    "ltr_enc_seed_set": (c_int, [c_uint64, P]),
    "ltr_enc_seed_advance": (c_int, [c_uint64, P]),
    "ltr_enc_seed_get": (c_int, [ctypes.POINTER(c_uint64)]),
    "ltr_enc_dropout_mask": (c_int, [c_uint64, c_int, c_int64, c_float, P, P]),
    "ltr_enc_attn_dropout_mask": (c_int, [c_uint64, c_int, c_int, c_int, c_int, c_float, P, P]),
    "ltr_enc_sum_partials": (c_int, [P, c_int, c_int64, c_int, P, P]),
    "ltr_enc_sum_partials_batch": (c_int, [P, c_int, P]),
    "ltr_enc_layernorm_fwd": (c_int, [P, P, P, c_int64, c_int, c_float, c_int, P, P, P]),
    "ltr_enc_layernorm_bwd": (c_int, [P, P, P, c_int64, c_int, c_float, c_int, P, P, c_int, P]),
    "ltr_enc_gemm_bf16": (c_int, [P, P]),
    "ltr_enc_colsum_bf16": (c_int, [P, c_int64, c_int, P, c_int, P]),
    "ltr_enc_drop_cast_colsum": (c_int, [P, c_int64, c_int, c_float, c_uint64, c_int, P, P, c_int, P]),
    "ltr_enc_attention_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_int, P, P]),
    "ltr_enc_attention_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_int, P, P]),
    "ltr_enc_attention_fwd_lse": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_int, P, P, P]),
    "ltr_enc_attention_bwd_lse": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_int, P, P]),
    "ltr_enc_attention_probs": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, c_uint64, c_int, P, P]),
    "ltr_enc_ffn_supported": (c_int, [c_int, c_int]),
    "ltr_enc_ffn_fwd": (c_int, [P, P, P, P, P, P, c_int64, c_int, c_int, c_float, c_uint64, c_int, c_int, P, P]),
    "ltr_enc_ffn_bwd_x": (c_int, [P, P, P, P, P, c_int64, c_int, c_int, c_float, c_uint64, c_int, P, P]),
    "ltr_enc_ffn_bwd_w": (c_int, [P, P, P, P, P, c_int64, c_int, c_int, c_float, c_uint64, c_int, c_int, P, P, P, P]),
    "ltr_enc_score_fwd": (c_int, [P, P, P, P, P, c_int64, c_int, c_float, c_int, P, P]),
    "ltr_enc_score_bwd": (c_int, [P, P, P, P, P, c_int64, c_int, c_float, c_int, P, P, c_int, P]),
    "ltr_ordinal_num_blocks": (c_int64, [c_int64]),
    "ltr_ordinal_fwd_bwd": (c_int, [P, P, c_int64, c_int, c_float, P, P, P, P]),
}

_lib = None


def library_path():
    return _SO


def lib():
    """Load (once) and return the ctypes handle.  Raises LtrBuildError if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise LtrBuildError(
                f"{_SO} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' at the repo root).  There is no CPU fallback.")
        try:
            h = ctypes.CDLL(_SO)
        except OSError as e:  # pragma: no cover - depends on the box
            raise LtrBuildError(f"cannot load {_SO}: {e}") from e
        for name, (res, args) in _PROTOTYPES.items():
            try:
                fn = getattr(h, name)
            except AttributeError as e:
                raise LtrBuildError(f"{_SO} does not export {name}: stale build?") from e
            fn.restype = res
            fn.argtypes = args
        from . import _scorer_protos
        _scorer_protos.bind(h)
        if h.ltr_abi_version() != 1:
            raise LtrBuildError(f"{_SO}: ABI version {h.ltr_abi_version()} != 1")
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise LtrError(f"{what}: {lib().ltr_error_string(rc).decode()} (code {rc})")
