"""GPU tier: bench.py keeps the driver's contract — one JSON line on stdout with the agreed keys, for one process and for a
two-rank launch through torch.distributed.run (gloo here: RCCL refuses two ranks on one device, and the box has one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"}


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_process_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d)
    assert d["unit"] == "images/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "bf16" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None                      # the committed PMC figure belongs to the batch-512 workload only
    assert d["cpu_baseline"] is None                  # --no-cpu-baseline
    assert d["final_loss"] > 0


def test_two_rank_launch_over_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "32", "--steps", "2", "--warmup", "1",
           "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 32 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["cpu_baseline"] is None


def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher - the parent never touches the GPU, starts two fresh rank
    processes and relays rank 0's one JSON line (gloo: the box has one GPU and RCCL refuses two ranks on one device)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "32", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["dp"]["world_size"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["dp"]["collectives_per_step"] == 13 and d["dp"]["backend"] == "gloo"
    assert abs(d["value"] - 2 * 32 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["cpu_baseline"] is None


def test_plain_command_refuses_more_rccl_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2 and "needs %d GPUs" % n in r.stderr and not r.stdout.strip()


@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_forced_single_rank_rccl_group(payload):
    """`--force-dp` initialises a ONE-rank `nccl` (= RCCL) process group on the one GPU there is and drives the
    engine's GradBucketReducer through it - init, 13 asynchronous collectives per step over the whole gradient, stream ordering and
    finish() on the real backend.  A one-rank all-reduce returns its input (the bf16 payload rounds every gradient to bf16 once on the
    way), so the run trains like the plain one."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-augment"]
    plain = subprocess.run(base, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-2000:]
    r = subprocess.run(base + ["--force-dp", "--grad-payload", payload], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d, p = _json_line(r.stdout), _json_line(plain.stdout)
    dp = d["dp"]
    assert dp["backend"] == "nccl (RCCL)" and dp["world_size"] == 1 and dp["forced_single_rank_group"] is True and dp["payload"] == payload
    assert dp["collectives_per_step"] == 13
    assert dp["allreduce_bytes_per_step"] == dp["gradient_bytes"] // (1 if payload == "fp32" else 2)
    assert dp["reducer_wait_ms_per_step"] >= 0 and dp["ms_per_step_without_exchange"] > 0
    assert p["dp"]["backend"] is None and p["dp"]["collectives_per_step"] == 0
    # two runs of the same step already differ in the last bits (fp32 atomics in the small weight-gradient reductions, amplified by
    # Adam's first steps): "the same training" is agreement of the loss after four steps to 1e-3, for either payload
    assert abs(d["final_loss"] - p["final_loss"]) < 1e-3 * abs(p["final_loss"])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1"
