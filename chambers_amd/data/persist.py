"""Whole-dataset save / load (reference: chambers/data/persist.py).

The reference writes `dataset.metadata` (element spec as JSON: shape, DataType enum, name per tensor; whether the elements were
enumerated for sharding) next to a `tf.data.experimental.save` snapshot.  The metadata file is the same here; the payload is
`n_files` TFRecord shards (`shard-00000-of-0000N.tfrecord`, element i in shard i % n_files, the reference's shard function)
written by `chambers_amd.data.tf_record` instead of TensorFlow's snapshot format, which has no public specification.
`load_dataset` restores the original element order across the shards.
"""
import json
import os

import numpy as np

from .dataset import Dataset
from .tf_record import _DT, DT_STRING, TFRecordWriter, make_dataset_deserialize_fn, serialize_to_example, tfrecord_iterator


def _tensor_spec_to_dict(t):
    if isinstance(t, (bytes, str)):
        return {"shape": [], "dtype": DT_STRING, "name": None}
    a = np.asarray(t)
    return {"shape": list(a.shape), "dtype": _DT[a.dtype], "name": None}


def _element_spec_to_json(element):
    if isinstance(element, tuple):
        return [_element_spec_to_json(e) for e in element]
    return _tensor_spec_to_dict(element)


def _dump_dataset_metadata(path, element_spec, enumerated, n_files, n_elements):
    with open(path, "w") as f:
        json.dump({"element_spec": element_spec, "enumerated": enumerated, "n_files": n_files, "n_elements": n_elements}, f)


def _load_dataset_metadata(path):
    with open(path, "r") as f:
        return json.load(f)


def _shard_name(path, k, n):
    return os.path.join(path, "shard-%05d-of-%05d.tfrecord" % (k, n))


def save_dataset(dataset, path, n_files=1):
    """persist.py:63-82.  The element spec recorded is that of the first element (shapes may differ between elements)."""
    if n_files < 1:
        raise ValueError("n_files must be >= 1")
    os.makedirs(path, exist_ok=True)
    writers = [TFRecordWriter(_shard_name(path, k, n_files)) for k in range(n_files)]
    spec, count = None, 0
    try:
        for i, e in enumerate(dataset):
            e = e if isinstance(e, tuple) else (e,)
            if spec is None:
                spec = _element_spec_to_json(e if len(e) > 1 else e[0])
            writers[i % n_files].write(serialize_to_example(*e))
            count += 1
    finally:
        for w in writers:
            w.close()
    _dump_dataset_metadata(os.path.join(path, "dataset.metadata"), spec, n_files > 1, n_files, count)


def load_dataset(path):
    """persist.py:85-92: the saved elements, in the order they were saved."""
    meta = _load_dataset_metadata(os.path.join(path, "dataset.metadata"))
    n = int(meta.get("n_files", 1))
    shards = [_shard_name(path, k, n) for k in range(n)]

    def raw():
        its = [tfrecord_iterator(p) for p in shards]
        k = 0
        while True:                               # element i lives in shard i % n
            try:
                yield (next(its[k % n]),)
            except StopIteration:
                return
            k += 1

    records = Dataset(raw)
    if meta.get("n_elements", 0) == 0:
        return records.take(0)
    fn = make_dataset_deserialize_fn(records)

    def gen():
        for (rec,) in records:
            out = fn(rec)
            yield out if isinstance(out, tuple) else (out,)

    return Dataset(gen)
