"""CPU tier: bench.py's own launcher (`python bench.py --gpus N` with no torch.distributed.run around it).  No GPU here, so the
rank processes cannot run the step; what is checked is the launcher's behaviour around them: the clear refusal when RCCL ranks
outnumber GPUs, a failing rank ending the job with a non-zero code, the timeout killing every rank it started."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def test_more_rccl_ranks_than_gpus_is_one_clear_line():
    import torch
    n = torch.cuda.device_count() + 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env=ENV)
    assert r.returncode == 2
    assert r.stderr.strip().count("\n") == 0 and "needs %d GPUs" % n in r.stderr
    assert not r.stdout.strip()


def test_failing_rank_fails_the_job():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("on a GPU box the ranks run; tests/test_bench_contract_gpu.py covers that")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=ENV)
    assert r.returncode not in (0, 2), r.stderr[-1500:]
    assert "stopping the other ranks" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_timeout_kills_every_rank(tmp_path):
    """The launcher with ranks that never finish (a stand-in script that sleeps): CHB_BENCH_TIMEOUT ends the job with 124 and no
    rank process survives."""
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "sleepy.py"
    stub.write_text("import os, time\nopen(os.path.join(%r, 'pid.' + os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(600)\n" % str(tmp_path))
    code = ("import sys, types; sys.path.insert(0, %r); import bench; bench.__file__ = %r; sys.argv = ['bench.py']\n"
            "sys.exit(bench.spawn_ranks(types.SimpleNamespace(gpus=3, backend='gloo')))\n" % (ROOT, str(stub)))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(ENV, CHB_BENCH_TIMEOUT="3"))
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert time.time() - t0 < 120
    pids = [int((tmp_path / ("pid.%d" % k)).read_text()) for k in range(3)]
    for pid in pids:
        assert not os.path.exists("/proc/%d" % pid) or open("/proc/%d/stat" % pid).read().split()[2] == "Z"
