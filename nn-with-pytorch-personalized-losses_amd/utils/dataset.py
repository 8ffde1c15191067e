"""MI355X drop-in for utils/dataset.py of the reference: LETOR / svmlight loader and the baseline-score CSV files
(svmDataset :9-30, get_data :33-69, load_baseline_file :72-101, store_baseline_data :104-138, get_baseline_data
:141-151), same names and signatures.

`get_data` parses with the native multi-threaded parser (csrc/ltr_data.hip; no sklearn, no per-document Python loop,
no joblib cache) and returns the query-grouped data as packed arrays -- X [Q, S, F] float32 and labels [Q, S] float64
(svmlight labels are float64, as in the reference) -- whenever every query has the same number of documents, which is
what the reference's drivers require (`torch.tensor(X_train)`, main_batch_execution.py:79); `len()`, indexing and
`torch.tensor(...)` behave as they do on the reference's list-of-lists.  Ragged collections come back as lists of
per-query arrays.  The per-epoch shuffle of the loaded tensors runs on the device: ltr_mi355x.data.EpochShuffler."""
import os

import numpy as np

from ltr_mi355x import data as _data


class svmDataset():
    def __init__(self, dataset_name):
        self.train_data_path = "D:\\Colecoes\\BD\\2003_td_dataset\\Fold1\\train.txt"
        self.test_data_path = "D:\\Colecoes\\BD\\2003_td_dataset\\Fold1\\test.txt"
        self.vali_data_path = "D:\\Colecoes\\BD\\2003_td_dataset\\Fold1\\vali.txt"

        self.baseline_train_data_path = "D:\\Colecoes\\BD\\2003_td_dataset\\Fold1\\baseline.train.txt"
        self.baseline_test_data_path = "D:\\Colecoes\\BD\\2003_td_dataset\\Fold1\\baseline.test.txt"
        self.baseline_vali_data_path = "D:\\Colecoes\\BD\\2003_td_dataset\\Fold1\\baseline.vali.txt"

        self.docs_per_query = 1000
        self.queries_on_test = 10
        self.queries_on_train = 10

        if "2003_td" in dataset_name:
            self.num_features = 64
            self.normalized_num_docs = True
        elif "web10k" in dataset_name:
            self.num_features = 136
            self.normalized_num_docs = True


def _path(info_dataset, type_file, prefix=""):
    # an unknown type_file leaves `data` unassigned in the reference (:37-42): same failure mode
    if type_file not in ("train", "test", "vali"):
        raise UnboundLocalError("local variable 'data' referenced before assignment")
    return getattr(info_dataset, f"{prefix}{type_file}_data_path")


def get_data(info_dataset, type_file="train"):
    """(features_docs_by_query, labels_by_query): documents grouped into queries where qid changes."""
    X, y, qid = _data.load_svmlight(_path(info_dataset, type_file))
    b = _data.query_bounds(qid)
    return _data.group_by_query(X, b), _data.group_by_query(y, b)


def load_baseline_file(name_file):
    """Lines `qid,score_1,...,score_n` -> baselines grouped by query: [Q, S, n] float32 (or a ragged list)."""
    raw = np.loadtxt(name_file, delimiter=",", dtype=np.float64, ndmin=2)
    qid = raw[:, 0].astype(np.int64)
    return _data.group_by_query(raw[:, 1:].astype(np.float32), _data.query_bounds(qid))


def store_baseline_data(name_method, info_dataset, queries_ids, predicted_values, append=True, type_file="train"):
    assert len(queries_ids) == len(predicted_values)
    name_file = _path(info_dataset, type_file, prefix="baseline_")
    info_file = name_file.replace("baseline", "baseline.info")
    if append and not os.path.isfile(name_file):
        append = False
    if append:
        with open(name_file, "r") as f:
            data = f.readlines()
        assert len(queries_ids) == len(data)
        with open(info_file, "a") as f:
            f.write(name_method + "\n")
        lines = [f"{data[i].rstrip(chr(10))},{round(predicted_values[i], ndigits=5)}\n" for i in range(len(queries_ids))]
    else:
        lines = [f"{queries_ids[i]},{round(predicted_values[i], ndigits=5)}\n" for i in range(len(queries_ids))]
    with open(name_file, "w") as f:
        f.writelines(lines)
    if not append:
        with open(info_file, "w") as f:
            f.write(name_method + "\n")


def get_baseline_data(info_dataset, type_file="train"):
    return load_baseline_file(_path(info_dataset, type_file, prefix="baseline_"))
