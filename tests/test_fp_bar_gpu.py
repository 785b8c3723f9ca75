"""The stated floating-point bar (SURVEY 8c / north star "within 1e-3 rel for bf16 attention / FFN"), asserted per tensor.

Every tensor family of one forward + loss + backward of the four BASELINE geometries (width, heads, tokens as configured; depth
cut to 2 blocks so the CPU oracle finishes in seconds) against the plain fp32 oracle - bounded by the rounding-point BUDGET of
tests/fp_budget.py, not by "measured x 1.5" - and against the oracle that emulates the build's bf16 rounding points (tight measured
bounds) (tests/fp_bar.py prints the table; the full-depth table is in DESIGN.md section 2).  What the numbers mean: a tensor the build stores in bf16 sits at least one storage rounding
(rel-L2 1.65e-3 on such data) from the fp32 oracle; k independent roundings along a path add up like a random walk,
~1.65e-3 * sqrt(k) - logits after 2 blocks (k ~ 12) 5.9e-3, gradients ~7e-3.  The kernels themselves add nothing to the storage
format: tests/test_vit_kernels_gpu.py holds every kernel with an fp32 output to <= 2e-6 of the fp64 result on the same bf16
operands and every bf16 output to bit-equality with the fp32 output rounded once.
"""
import pytest

from conftest import fp_check
from fp_bar import GEOMETRIES, measure
from fp_budget import FLOOR, budget_for_row

pytestmark = pytest.mark.gpu

# vs the bf16-EMULATING oracle (same rounding points: what is left are rounding flips and accumulation order) - the kernel-defect
# detector, tight: max over the four geometries of the depth-2 measurement x 1.5.  The column against the plain fp32 oracle is
# bounded by the rounding-point budget of tests/fp_budget.py instead (floor x sqrt(k + kappa^2) x 1.5, k counted, kappa computed).
EMU_BOUNDS = {
    "logits": 6.6e-3,
    "o (last block)": 1.8e-3,
    "o (block 0)": 8.6e-4,
    "loss per sample": 1.3e-3,
    "dq (block 0)": 1.33e-2,
    "dk (block 0)": 1.07e-2,
    "dv (block 0)": 1.02e-2,
    "dO (block 0)": 9.3e-3,
}
EMU_GRAD_BOUND = 1.15e-2             # every weight-gradient family ("d ..." rows)
# vs the plain fp32 oracle the bound is min(budget, CAP): the budget says what the rounding points allow, the cap (max over the four
# geometries of the depth-2 measurement x 1.5, profiles/r03_fp_bar_depth2.txt) makes a 1.5x regression of ANY tensor fail even
# where the budget - whose kappa is computed from the oracle's own tensors at run time - has room (VERDICT r3 item 5 / ADVICE r3: dq
# measures 1.7-2.0e-2 under a budget of 3.6-5.2e-2; loss per sample 1.2e-3 under 8.6e-3).
FP32_CAPS = {
    "logits": 9.3e-3,
    "o (last block)": 5.3e-3,
    "o (block 0)": 3.7e-3,
    "loss per sample": 1.9e-3,
    "dk (block 0)": 1.37e-2,
    "dv (block 0)": 1.12e-2,
    "dO (block 0)": 1.06e-2,
}
FP32_GRAD_CAP = 1.21e-2              # every weight-gradient family
FP32_DQ_CAPS = {"config3": 2.52e-2, "config4": 2.90e-2, "config5": 3.01e-2}      # dq: delta is formed from the stored bf16 O
KAPPA_MAX = 25.0                     # the computed cancellation ratio of dq (13.5 / 16.2 / 20.3 at the three training geometries)
FLOOR_RANGE = (1.5e-3, 1.8e-3)       # one bf16 storage rounding of such tensors (rel-L2); the model's FLOOR sits inside


@pytest.mark.parametrize("geometry", list(GEOMETRIES))
def test_fp_bar(geometry):
    rows = measure(geometry, depth=2)
    depth = rows.pop("_depth")
    assert depth == 2 and "logits" in rows and "o (last block)" in rows and FLOOR_RANGE[0] < FLOOR < FLOOR_RANGE[1]
    for row, v in rows.items():
        b32 = budget_for_row(row, depth, v.get("kappa"))
        if row.startswith("dq"):
            cap = [c for g, c in FP32_DQ_CAPS.items() if geometry.startswith(g)]
            assert cap and v.get("kappa") and v["kappa"] < KAPPA_MAX, (geometry, v.get("kappa"))
            b32 = min(b32, cap[0])
        else:
            b32 = min(b32, FP32_CAPS.get(row, FP32_GRAD_CAP))
        fp_check("%s | %s | vs fp32 (budget: k, kappa = %s)" % (geometry, row, "%.1f" % v["kappa"] if v.get("kappa") else "-"), v["fp32"], b32)
        fp_check("%s | %s | vs bf16-emu" % (geometry, row), v["emu"], EMU_BOUNDS.get(row, EMU_GRAD_BOUND))
        assert v["emu"] <= v["fp32"] * 1.05 or v["emu"] < 1e-3       # sharing the rounding points can only bring the oracle closer
        if v["floor"] is not None:
            assert FLOOR_RANGE[0] < v["floor"] < FLOOR_RANGE[1], (row, v["floor"])
            assert v["fp32"] > 0.5 * v["floor"]          # a bf16-stored tensor cannot be (much) closer to fp32 than one rounding
