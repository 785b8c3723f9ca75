import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for both tiers (CPU tier only loads it and checks symbols)."""
    from chambers_amd import _build
    if not _build.is_current():
        _build.build(verbose=False)
    yield


# ---- measured floating-point errors ------------------------------------------------------------------------------------
# GPU parity tests call fp_check(name, measured, bound): the assertion is measured < bound, and the measured value is recorded
# so that bounds can be stated as "measured x 1.5" (the values land in gpurun_out/fp_measured.json on the GPU box; the table in
# DESIGN.md section 2 comes from tests/fp_bar.py).
_FP_MEASURED = {}


def fp_check(name, measured, bound):
    measured = float(measured)
    prev = _FP_MEASURED.get(name)
    _FP_MEASURED[name] = {"measured": max(measured, prev["measured"]) if prev else measured, "bound": float(bound)}
    assert measured < bound, "%s: measured %.3e, bound %.3e" % (name, measured, bound)


def pytest_sessionfinish(session, exitstatus):
    out_dir = os.path.join(ROOT, "gpurun_out")
    if _FP_MEASURED and os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "fp_measured.json"), "w") as f:
            json.dump(_FP_MEASURED, f, indent=1, sort_keys=True)
