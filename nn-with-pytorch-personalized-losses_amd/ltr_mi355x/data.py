"""Host side of the data path: LETOR / svmlight files -> packed arrays (native multi-threaded
parser, csrc/ltr_data.hip) and the per-epoch shuffle on the device (ltr_gather_rows_f32)."""
import ctypes
import os

import numpy as np
import torch

from ._lib import LtrError, check, lib
from .functional import _ptr, _stream, require_device


def load_svmlight(path, n_features=None, zero_based="auto", n_threads=0):
    """Parse a LETOR / svmlight file in file order.  Returns (X [n_docs, F] float32, y [n_docs] float64,
    qid [n_docs] int64).  F defaults to what sklearn's load_svmlight_file infers (the largest feature id, ids
    zero-based iff a 0 id occurs -- `zero_based="auto"`)."""
    h = lib()
    n_docs, lo, hi = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
    bpath = os.fsencode(path)
    check(h.ltr_svmlight_scan(bpath, ctypes.byref(n_docs), ctypes.byref(lo), ctypes.byref(hi), int(n_threads)),
          f"ltr_svmlight_scan({path})")
    base = (0 if lo.value == 0 else 1) if zero_based == "auto" else (0 if zero_based else 1)
    inferred = max(hi.value - base + 1, 0)
    F = int(n_features) if n_features else inferred
    if F < 1:
        raise LtrError(f"{path}: no features found")
    if inferred > F:
        raise ValueError(f"{path}: feature id {hi.value} does not fit n_features={F}")
    X = np.empty((n_docs.value, F), dtype=np.float32)
    y = np.empty(n_docs.value, dtype=np.float64)
    qid = np.empty(n_docs.value, dtype=np.int64)
    if n_docs.value:
        check(h.ltr_svmlight_load(bpath, n_docs.value, F, base, X.ctypes.data, y.ctypes.data, qid.ctypes.data, int(n_threads)),
              f"ltr_svmlight_load({path})")
    return X, y, qid


def query_bounds(qid):
    """Start offsets of the runs of equal qid (a new query starts where qid changes, utils/dataset.py:54-60),
    with the total appended: query q is rows [b[q], b[q+1])."""
    qid = np.asarray(qid)
    if qid.size == 0:
        return np.zeros(1, dtype=np.int64)
    starts = np.flatnonzero(np.concatenate(([True], qid[1:] != qid[:-1])))
    return np.concatenate((starts, [qid.size])).astype(np.int64)


def group_by_query(rows, bounds):
    """rows [n_docs, ...] -> [Q, S, ...] when every query has the same number of documents (the reference's
    `normalized_num_docs` collections, utils/dataset.py:25,30), else a list of per-query arrays (ragged)."""
    sizes = np.diff(bounds)
    if sizes.size and (sizes == sizes[0]).all():
        return rows.reshape((sizes.size, int(sizes[0])) + rows.shape[1:])
    return [rows[bounds[q]:bounds[q + 1]] for q in range(sizes.size)]


def gather_rows(src, idx, out=None):
    """out[r] = src[idx[r]] over the leading dimension, on the device (the per-epoch `X_train[idx]` of
    main_batch_execution.py:112-117).  src fp32 device tensor, idx int64 device tensor.  Negative indices count from the end as
    in torch indexing; an index that is still out of range gives a zero row (torch raises a device-side assert)."""
    require_device(src, idx)
    if src.dtype != torch.float32:
        raise TypeError(f"gather_rows moves fp32 rows, got {src.dtype}")
    s = src.detach().contiguous()
    ii = idx.to(torch.int64).contiguous()
    n = int(ii.numel())
    row = 1
    for d in s.shape[1:]:
        row *= int(d)
    if out is None:
        out = torch.empty((n,) + tuple(s.shape[1:]), dtype=torch.float32, device=s.device)
    elif out.shape != (n,) + tuple(s.shape[1:]) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 tensor of shape [len(idx), ...]")
    if out.data_ptr() == s.data_ptr() and n:
        raise ValueError("gather_rows cannot permute in place")
    if n == 0:
        return out
    with torch.cuda.device(s.device):
        check(lib().ltr_gather_rows_f32(_ptr(s), int(s.shape[0]), _ptr(ii), n, max(row, 1), _ptr(out), _stream()),
              "ltr_gather_rows_f32")
    return out


class EpochShuffler:
    """The reference's per-epoch `idx = torch.randperm(Q); X, y, yb = X[idx], y[idx], yb[idx]` with two resident buffers
    per tensor that swap roles every epoch (a full-tensor gather cannot run in place): no allocation after the first
    epoch, one HBM-bound launch per tensor."""

    def __init__(self, *tensors):
        require_device(*tensors)
        self.cur = [t.detach().to(torch.float32).contiguous() for t in tensors]
        self.alt = [torch.empty_like(t) for t in self.cur]

    def shuffle(self, generator=None):
        Q = self.cur[0].shape[0]
        idx = torch.randperm(Q, device=self.cur[0].device, generator=generator)
        for i in range(len(self.cur)):
            gather_rows(self.cur[i], idx, out=self.alt[i])
        self.cur, self.alt = self.alt, self.cur
        return idx, tuple(self.cur)
