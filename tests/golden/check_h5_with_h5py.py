"""Reads an HDF5 file with REAL h5py / libhdf5 and prints what it finds as one JSON line - the cross-check of
chambers_amd.utils.hdf5_lite's WRITER (tests/test_hdf5_lite.py runs it when an interpreter with h5py exists on the box; this
image has /opt/conda/bin/python3.9 with h5py 3.3.0 / libhdf5 1.10.6; the test-suite interpreter has none).

    /opt/conda/bin/python3.9 tests/golden/check_h5_with_h5py.py file.h5
"""
import hashlib
import json
import sys

import h5py
import numpy as np


def _attr(v):
    if isinstance(v, bytes):
        return v.decode()
    a = np.asarray(v)
    if a.dtype.kind in "SO":
        return [x.decode() if isinstance(x, bytes) else str(x) for x in a.reshape(-1)]
    return a.tolist()


def main(path):
    out = {"attrs": {}, "datasets": {}, "groups": []}
    with h5py.File(path, "r") as f:
        out["attrs"]["/"] = {k: _attr(v) for k, v in f.attrs.items()}

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                a = np.ascontiguousarray(obj[()])
                out["datasets"][name] = {"shape": list(obj.shape), "dtype": str(obj.dtype), "sha1": hashlib.sha1(a.tobytes()).hexdigest()}
            else:
                out["groups"].append(name)
                if len(obj.attrs):
                    out["attrs"][name] = {k: _attr(v) for k, v in obj.attrs.items()}
        f.visititems(visit)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
