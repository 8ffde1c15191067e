"""CPU: the native LETOR / svmlight parser (host code of the C-ABI library, no GPU involved) against sklearn's
load_svmlight_file -- the parser the reference calls (utils/dataset.py:37-42) -- on synthetic files, and the
reference-shaped wrappers of utils/dataset.py (query grouping, baseline CSV round trip)."""
import os

import numpy as np
import pytest


def _write_letor(path, Q, S, F, seed, one_based=True, sparse=0.3, comments=True, ragged=False):
    rng = np.random.default_rng(seed)
    rows = []
    with open(path, "w") as f:
        if comments:
            f.write("# a header comment line\n\n")
        for q in range(Q):
            n = S if not ragged else int(rng.integers(1, S + 1))
            for d in range(n):
                y = int(rng.integers(0, 5))
                x = rng.standard_normal(F).round(6)
                x[rng.random(F) < sparse] = 0.0
                feats = " ".join(f"{j + (1 if one_based else 0)}:{float(x[j])!r}" for j in range(F) if x[j] != 0.0)
                if not one_based and x[0] == 0.0:       # keep a 0 id present so "auto" detects zero-based files
                    feats = f"0:0 {feats}"
                tail = f" #docid = GX{q:03d}-{d}" if comments and d % 3 == 0 else ""
                f.write(f"{y} qid:{q + 10} {feats}{tail}\n")
                rows.append((q + 10, y, x))
    return rows


@pytest.mark.parametrize("one_based", [True, False])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_parser_matches_sklearn(tmp_path, one_based, threads):
    from sklearn.datasets import load_svmlight_file
    from ltr_mi355x import data
    p = str(tmp_path / "train.txt")
    _write_letor(p, Q=23, S=17, F=46, seed=1 + threads, one_based=one_based)
    X, y, qid = data.load_svmlight(p, n_threads=threads)
    ref = load_svmlight_file(p, query_id=True)
    assert X.dtype == np.float32 and y.dtype == np.float64 and qid.dtype == np.int64
    assert np.array_equal(X, ref[0].toarray().astype(np.float32))
    assert np.array_equal(y, ref[1]) and np.array_equal(qid, ref[2])


def test_get_data_groups_queries_like_the_reference(tmp_path):
    """features_docs_by_query / labels_by_query: [Q, S, F] float32 and [Q, S] float64, torch.tensor()-able, len() = Q."""
    import torch
    from utils.dataset import get_baseline_data, get_data, store_baseline_data, svmDataset
    info = svmDataset("web10k")
    assert info.num_features == 136 and info.normalized_num_docs
    info.train_data_path = str(tmp_path / "Norm.train.txt")
    info.baseline_train_data_path = str(tmp_path / "baseline.Norm.train.txt")
    rows = _write_letor(info.train_data_path, Q=9, S=12, F=136, seed=5)
    X, y = get_data(info, "train")
    assert len(X) == 9 and len(y) == 9
    Xt, yt = torch.tensor(X), torch.tensor(y)
    assert Xt.shape == (9, 12, 136) and Xt.dtype == torch.float32 and yt.shape == (9, 12) and yt.dtype == torch.float64
    assert np.array_equal(X.reshape(-1, 136), np.stack([r[2] for r in rows]).astype(np.float32))
    assert np.array_equal(y.reshape(-1), np.array([r[1] for r in rows], dtype=np.float64))
    # baseline CSV: write one system, append a second, read back grouped by query
    qids = [r[0] for r in rows]
    s1 = [float(i % 7) / 3 for i in range(len(rows))]
    s2 = [float(i % 5) - 1.23456789 for i in range(len(rows))]
    store_baseline_data("linear", info, qids, s1, append=True, type_file="train")      # no file yet -> plain write
    store_baseline_data("tree", info, qids, s2, append=True, type_file="train")
    yb = get_baseline_data(info, "train")
    assert yb.shape == (9, 12, 2) and yb.dtype == np.float32
    assert np.allclose(yb[:, :, 0].reshape(-1), np.round(s1, 5)) and np.allclose(yb[:, :, 1].reshape(-1), np.round(s2, 5))
    assert open(info.baseline_train_data_path.replace("baseline", "baseline.info")).read().split() == ["linear", "tree"]
    with pytest.raises(UnboundLocalError):
        get_data(info, "nope")


def test_ragged_queries_and_errors(tmp_path):
    from ltr_mi355x import LtrError, data
    p = str(tmp_path / "ragged.txt")
    rows = _write_letor(p, Q=6, S=9, F=20, seed=3, ragged=True)
    X, y, qid = data.load_svmlight(p, n_features=20)
    b = data.query_bounds(qid)
    groups = data.group_by_query(X, b)
    assert isinstance(groups, list) and len(groups) == 6 and sum(len(g) for g in groups) == len(rows)
    with pytest.raises(LtrError, match="cannot open"):
        data.load_svmlight(str(tmp_path / "missing.txt"))
    bad = str(tmp_path / "bad.txt")
    open(bad, "w").write("1 qid:1 1:0.5 oops\n")
    with pytest.raises(LtrError, match="malformed"):
        data.load_svmlight(bad)
    with pytest.raises(ValueError):
        data.load_svmlight(p, n_features=5)
    empty = str(tmp_path / "empty.txt")
    open(empty, "w").write("# nothing\n")
    with pytest.raises(LtrError):
        data.load_svmlight(empty)


def test_parser_never_reads_past_a_line(tmp_path):
    """Every numeric parse is bounded by the line end (the mapping is not NUL-terminated and strtod skips '\\n'):
    an empty value ("2:" at the end of a line) is malformed -- it must NOT swallow the next line's label --, so are
    non-finite and over-long values; a file whose size is an exact multiple of the page size and that lacks a final
    newline parses without touching the byte behind the mapping."""
    import mmap
    from ltr_mi355x import LtrError, data
    for k, text in enumerate(["1 qid:1 1:0.5 2:\n7 qid:1 1:0.25\n", "1 qid:1 1:0.5 2:", "1 qid:1 1:nan\n", "1 qid:1 1:1e400\n",
                              "1 qid:1 1:inf\n", "1 qid: 1:0.5\n", "1 qid:1 :0.5\n", "1 qid:1 1:" + "9" * 80 + "\n", "\n \n1e\n",
                              "1 qid:1 1:0.5 2 :3\n", "1 qid:1 1:0.5x\n"]):
        bad = str(tmp_path / f"bad{k}.txt")
        with open(bad, "w") as f:
            f.write(text)
        with pytest.raises(LtrError, match="malformed"):
            data.load_svmlight(bad)
    page = mmap.PAGESIZE
    line = "2 qid:7 1:0.5 2:-1.25 3:3\n"
    last = "1 qid:8 1:0.125 3:"
    n_lines = (2 * page - len(last) - 1) // len(line)
    body = line * n_lines
    fill = 2 * page - len(body) - len(last)
    assert 1 <= fill < 60
    text = body + last + "7" * fill            # last token ends exactly at the end of the mapping, no newline
    assert len(text) == 2 * page
    p = str(tmp_path / "page.txt")
    with open(p, "w") as f:
        f.write(text)
    X, y, qid = data.load_svmlight(p, n_features=3, n_threads=3)
    assert X.shape == (n_lines + 1, 3)
    assert np.array_equal(X[0], np.float32([0.5, -1.25, 3.0])) and y[0] == 2 and qid[0] == 7
    assert np.array_equal(X[-1], np.float32([0.125, 0.0, float("7" * fill)])) and y[-1] == 1 and qid[-1] == 8
    # accepted spellings stay accepted: exponents, signs, leading '+', CRLF line ends, trailing comment
    ok = str(tmp_path / "ok.txt")
    with open(ok, "w") as f:
        f.write("+1.5e0 qid:3 1:-2.5E-1 2:+4 # c\r\n0 qid:3 2:.5\r\n")
    X, y, qid = data.load_svmlight(ok, n_features=2)
    assert np.array_equal(X, np.float32([[-0.25, 4.0], [0.0, 0.5]])) and list(y) == [1.5, 0.0] and list(qid) == [3, 3]
