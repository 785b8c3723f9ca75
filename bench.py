#!/usr/bin/env python
"""Headline benchmark: images/sec of a ViT-B/16 224^2 TRAINING step on synthetic ImageNet-shaped batches
(BASELINE.json `metric`; workload = configs[2]: on-GPU RandAugment(n=2, m=9) -> normalise -> forward ->
softmax-CE -> backward -> AdamW, batch 512 per GPU, bf16 MFMA compute / fp32 master weights).

    python bench.py --gpus N --steps K --warmup W          (N = 1..8; for N > 1 the command starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU, weak scaling (512 images per GPU), one logical gradient all-reduce per step over RCCL
(bucketed, overlapped with backward).  Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel
(the bf16 MFMA GEMM), measured live with HIP events on the launch stream inside the timed region;
`cpu_baseline` is the CPU oracle (oracle/, kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)

MODELS = {
    "vitti16": dict(patch_size=16, patch_dim=192, n_encoder_layers=12, n_heads=3, ff_dim=768),
    "vits16": dict(patch_size=16, patch_dim=384, n_encoder_layers=12, n_heads=6, ff_dim=1536),
    "vitb16": dict(patch_size=16, patch_dim=768, n_encoder_layers=12, n_heads=12, ff_dim=3072),
    "vitl16": dict(patch_size=16, patch_dim=1024, n_encoder_layers=24, n_heads=16, ff_dim=4096),
}


def forward_flops_per_image(cfg):
    """2*n*(3p^2)*D + L*(8*N*D^2 + 4*N^2*D + 4*N*D*ff) + 2*D*classes."""
    n, N, D, L, ff = cfg.n_patches, cfg.n_tokens, cfg.patch_dim, cfg.n_encoder_layers, cfg.ff_dim
    return 2 * n * cfg.patch_k * D + L * (8 * N * D * D + 4 * N * N * D + 4 * N * D * ff) + 2 * D * cfg.classes


class KernelTimer:
    """The library's launch profiler (chb_profile_enable / chb_profile_collect, include/chambers_hip.h): HIP events recorded by
    chb_gemm_nt / chb_gemm_tn_ws themselves, on the stream each GEMM is launched on, right before and right after the launch - also
    when the launch comes from inside a block-level call (chb_vit_block_fwd / _bwd), where no Python code runs between two GEMMs."""

    NT = {1: "gemm_nt_kernel", 2: "gemm_nt256_kernel", 3: "gemm_nt128_kernel", 4: "gemm_nt256pp_kernel", 5: "gemm_nt256sp_kernel"}
    TN = {0: "gemm_tn_kernel", 1: "gemm_tn256_kernel"}

    def __init__(self):
        self.records = []   # (name, work, start_ms, ms)
        self.enabled = False

    def start(self):
        from chambers_amd import _lib
        _lib.profile_enable(True)
        self.enabled = True

    def stop(self):
        """Call after the device has been synchronised."""
        from chambers_amd import _lib
        _lib.profile_enable(False)
        self.enabled = False
        self.records = []
        for r in _lib.profile_collect():
            if r["kind"] == 0:
                name = "%s<%d, %d>" % (self.NT.get(r["family"], "gemm_nt?"), r["epilogue"], r["out_dtype"])
            else:
                name = self.TN.get(r["family"], "gemm_tn?")
            self.records.append((name, 2.0 * r["m"] * r["n"] * r["k"], r["start_ms"], r["ms"]))

    def summary(self):
        agg = {}
        for name, work, _t0, ms in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += work
        return {k: {"launches": v[0], "total_ms": v[1], "avg_us": 1e3 * v[1] / v[0], "tflops": v[2] / (v[1] * 1e-3) / 1e12,
                    "work": v[2]} for k, v in agg.items()}

    def union_ms(self):
        """Wall time during which at least one timed launch was running: launches of two streams overlap (the engine runs
        weight-gradient GEMMs beside the dgrad chain), so the sum of durations counts shared time twice."""
        if not self.records:
            return 0.0
        iv = sorted((t0, t0 + ms) for _n, _w, t0, ms in self.records)
        total, cur_a, cur_b = 0.0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > cur_b:
                total += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        return total + (cur_b - cur_a)

    @staticmethod
    def families(ks):
        """Template instantiations of one kernel (gemm_nt256sp_kernel<EPI, OUT>) belong to one family: the dominant kernel of
        the step is chosen among families, never among the instantiations of one of them."""
        fam = {}
        for k, v in ks.items():
            f = fam.setdefault(k.split("<")[0], {"launches": 0, "total_ms": 0.0, "work": 0.0})
            f["launches"] += v["launches"]
            f["total_ms"] += v["total_ms"]
            f["work"] += v["work"]
        for f in fam.values():
            f["avg_us"] = 1e3 * f["total_ms"] / f["launches"]
            f["tflops"] = f["work"] / (f["total_ms"] * 1e-3) / 1e12
        return fam


def draw_randaugment_decisions(gen, n_transforms, batch, h, w):
    return [{"op": int(gen.integers(0, 16)), "negate": bool(gen.uniform() < 0.5),
             "centers": np.stack([gen.integers(0, h, size=batch), gen.integers(0, w, size=batch)], axis=1).astype(np.int32)}
            for _ in range(n_transforms)]


def _median_time(fn, warmup, runs, what=""):
    ts = []
    for i in range(warmup + runs):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if i >= warmup:
            ts.append(dt)
        if what:       # a line per run on stderr: the CPU leg takes minutes and must not look hung
            print("[bench cpu_baseline] %s: %s %d/%d %.2f s" % (what, "warm-up" if i < warmup else "run", i + 1 if i < warmup else i + 1 - warmup,
                                                              warmup if i < warmup else runs, dt), file=sys.stderr, flush=True)
    return float(np.median(ts))


def host_cores():
    """The CPU share this process really has: the affinity mask, cut down to the cgroup's CPU quota when there is one and to 16
    otherwise-unlimited threads (a GPU box shows all 256 hardware threads of its host in the mask but gives a job a share of
    them; torch with 256 threads on these problem sizes is slower than with 1)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(cfg_kwargs, train_batch=32, runs=5, warmup=2):
    """SURVEY 8(d) / BASELINE.md 4 protocol, on this host: the oracle (fp32 torch-CPU / NumPy restatement of the TF2 reference,
    kind "port") timed as the median of `runs` runs after `warmup` warm-ups, at k = all host cores this process may use and k = 1:
      * the metric's workload: RandAugment(2, 9) + normalise + ViT-B/16 forward + CE + backward + AdamW on B = 32 images
        (decisions from seed 42, dropout 0.1) at k = all cores -> images/s (`value`);
      * config 1 (ViT-Ti/16 forward, uint8[8,224,224,3] -> logits) at k = 1 and k = all cores."""
    from chambers_amd.engine import ViTConfig, init_keras_weights
    from oracle import augment_ref as A
    from oracle import rng_ref, vit_ref
    cores = host_cores()
    g = np.random.Generator(np.random.PCG64(0))

    # ---- config 1: ViT-Ti/16 forward, batch 8
    ti = ViTConfig(**dict(MODELS["vitti16"], dropout_rate=0.1, image_size=(224, 224), classes=1000))
    ti_w = {k: torch.tensor(v, dtype=torch.float32) for k, v in init_keras_weights(ti, seed=1234).items()}
    ti_images = g.integers(0, 256, size=(8, 224, 224, 3), dtype=np.uint8)

    def ti_forward():
        with torch.no_grad():
            vit_ref.vit_forward(ti_w, torch.from_numpy(A.imagenet_normalize(ti_images, "tf")), ti.as_oracle_cfg(), keys=None)

    cfg1 = {}
    for k in (1, cores):
        torch.set_num_threads(k)
        t = _median_time(ti_forward, warmup, runs, what="config 1 ViT-Ti/16 forward B=8, %d thread(s)" % k)
        cfg1["k%d" % k] = {"threads": k, "images_per_sec": 8 / t, "ms": 1e3 * t}

    # ---- the metric's workload on the CPU: ViT-B/16 train step, B = 32, all cores
    torch.set_num_threads(cores)
    cfg = ViTConfig(**cfg_kwargs)
    kw = init_keras_weights(cfg, seed=1234)
    p = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in kw.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v_ = {k: torch.zeros_like(v) for k, v in p.items()}
    gd = np.random.Generator(np.random.PCG64(42))
    images = g.integers(0, 256, size=(train_batch,) + cfg.image_size + (3,), dtype=np.uint8)
    labels = torch.as_tensor(g.integers(0, cfg.classes, size=(train_batch,)))
    n_sites = 1 + 3 * cfg.n_encoder_layers
    state = {"step": 0}

    def train_step():
        step = state["step"]
        dec = draw_randaugment_decisions(gd, 2, train_batch, *cfg.image_size)
        xa = A.rand_augment(images, 2, 9, dec)
        x = torch.from_numpy(A.imagenet_normalize(xa, "tf"))
        # dropout masks from torch's own generator: the oracle's counter-hash masks (NumPy uint64, what parity tests use) cost
        # more than the model and are this build's definition, not work the reference does
        keys = {s: vit_ref.NATIVE_DROPOUT for s in range(n_sites)}
        logits = vit_ref.vit_forward(p, x, cfg.as_oracle_cfg(), keys=keys)
        loss = vit_ref.sparse_ce_from_logits(logits, labels)
        grads = torch.autograd.grad(loss, list(p.values()))
        with torch.no_grad():
            vit_ref.adamw_step({k: v.data for k, v in p.items()}, dict(zip(p.keys(), grads)), m, v_, step + 1, weight_decay=0.05)
        state["step"] = step + 1

    # keep the default bench within minutes on any host: one probe step first; a host that needs > 20 s for it gets 1 + 3 runs
    # instead of 2 + 5 (the probe counts as the first warm-up either way; `sample` says what was done)
    t0 = time.perf_counter()
    train_step()
    probe = time.perf_counter() - t0
    print("[bench cpu_baseline] ViT-B/16 train step B=%d, %d threads: warm-up 1 %.2f s" % (train_batch, cores, probe), file=sys.stderr, flush=True)
    if probe > 20.0:
        warmup, runs = 1, 3
    t = _median_time(train_step, warmup - 1, runs, what="ViT-B/16 train step B=%d, %d threads" % (train_batch, cores))
    return {"value": train_batch / t, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "CPU restatement of the TF2 reference (oracle/, fp32 torch-CPU + NumPy): the same train step (RandAugment(2,9) + "
                      "normalise + forward + CE + backward + AdamW, dropout 0.1 with torch-native masks) on B=%d images, median of %d runs after %d warm-ups, %d threads"
                      % (train_batch, runs, warmup, cores),
            "ms_per_step": 1e3 * t,
            "config1_vitti16_forward_b8": cfg1}


def pmc_traffic(kernel):
    """Per-launch HBM bytes of `kernel` from the newest profiles/*bench_pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_bench.sh); None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*bench_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        table = json.load(f).get("kernels", {})
    # the live timer groups a kernel's template instantiations that are picked per call (gemm_tn256_kernel<COLSUM, FAST>: with /
    # without the ride-along bias gradient; trailing FAST = staging without clamps, gemm_nt256_kernel<EPI, OUT, FAST>); rocprofv3
    # lists them separately: launch-weighted mean over them
    rows = [v for k, v in table.items() if k.split("<")[0] == kernel.split("<")[0]]
    n = sum(v["launches_sampled"] for v in rows)
    if not rows or not n:
        return None
    return {"hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in rows) / n,
            "source": "profiles/" + os.path.basename(files[-1])}


def ensure_library(local_rank):
    """The HIP library normally travels prebuilt with the tree (__graft_entry__.build()); if its stamp is stale or it is
    missing, local rank 0 compiles it (hipcc is on the GPU box) and the other ranks wait for the stamp."""
    from chambers_amd import _build
    if _build.is_current():
        return
    if local_rank == 0:
        _build.build(verbose=False)
        return
    deadline = time.time() + 900
    while not _build.is_current():
        if time.time() > deadline:
            raise RuntimeError("libchambers_hip.so was not built by local rank 0 within 15 minutes")
        time.sleep(2.0)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this same script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, one per GPU), relay their output (only rank 0 prints the JSON line), return non-zero when any rank
    fails or the job exceeds CHB_BENCH_TIMEOUT seconds (default 1800).  This parent makes no HIP call at any point
    (`torch.cuda.device_count()` does not initialise the GPU on this image) and never exec()s; a failed or overdue job is ended by
    killing exactly the process groups started here."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    if args.backend == "nccl" and torch.cuda.device_count() < n:
        print("bench.py: --gpus %d on backend nccl (RCCL) needs %d GPUs, this node shows %d (RCCL refuses two ranks on one device; "
              "--backend gloo rehearses N ranks on fewer GPUs)" % (n, n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    timeout = float(os.environ.get("CHB_BENCH_TIMEOUT", "1800"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0 inherits stdout (its one JSON line is the job's); the other ranks' stdout goes to stderr so nothing else can land there
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=os.getcwd(),
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)      # the group this launcher created for that rank, nothing else
                except ProcessLookupError:
                    pass
        for p in procs:
            p.wait()

    deadline = time.time() + timeout
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % bad[0], file=sys.stderr)
                rc = bad[0][1] if bad[0][1] > 0 else 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                print("bench.py: %d-rank job exceeded %.0f s (CHB_BENCH_TIMEOUT); killing it" % (n, timeout), file=sys.stderr)
                rc = 124
                break
            time.sleep(0.2)
    finally:
        kill_all()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU")
    ap.add_argument("--model", default="vitb16", choices=sorted(MODELS))
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-augment", action="store_true")
    ap.add_argument("--augment", default="randaugment", choices=("randaugment", "autoaugment"),
                    help="on-GPU augmentation stage: RandAugment(2, 9) (configs 3/4) or AutoAugment policy v0 (config 5)")
    ap.add_argument("--elementwise", action="store_true", help="per-image augmentation decisions (the schemes' elementwise=True mode)")
    ap.add_argument("--unfused-augment", action="store_true", help="batch-shared chain as one launch per op + a separate patchify pass")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to rehearse N ranks on one GPU")
    ap.add_argument("--force-dp", action="store_true",
                    help="N = 1 only: initialise a ONE-rank process group on the chosen backend and drive the gradient exchange through it "
                         "(every collective is issued and waited for; a one-rank all-reduce returns its input) - the RCCL code path on one GPU")
    ap.add_argument("--grad-payload", default=os.environ.get("CHB_GRAD_PAYLOAD", "fp32"), choices=("fp32", "bf16"),
                    help="dtype the gradient slices are all-reduced in (bf16: half the bytes, one extra rounding per contribution)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU) and the N ranks are
        # fresh children
        raise SystemExit(spawn_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if args.backend == "nccl" and world > 1 and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks on backend nccl (RCCL) need %d GPUs, this node shows %d (RCCL refuses two ranks on one "
                         "device; --backend gloo rehearses N ranks on fewer GPUs)" % (world, world, torch.cuda.device_count()))
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    ensure_library(int(os.environ.get("LOCAL_RANK", "0")))
    dist = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            sock.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from chambers_amd import augmentations as aug
    from chambers_amd import kernels as K
    from chambers_amd.engine import ViTConfig, ViTEngine, init_keras_weights

    timer = KernelTimer()

    cfg_kwargs = dict(MODELS[args.model], dropout_rate=0.1, image_size=(args.image_size, args.image_size), classes=1000)
    cfg = ViTConfig(**cfg_kwargs)
    eng = ViTEngine(cfg, args.batch, training=True, seed=rank, grad_payload=args.grad_payload,      # dropout masks differ per rank, like the data
                    force_dp=args.force_dp and world == 1)
    eng.load_keras_weights(init_keras_weights(cfg, seed=1234))      # same init on every rank
    g = np.random.Generator(np.random.PCG64(rank))                  # synthetic data: seed = rank
    images = torch.as_tensor(g.integers(0, 256, size=(args.batch, args.image_size, args.image_size, 3), dtype=np.uint8), device="cuda")
    labels = torch.as_tensor(g.integers(0, 1000, size=(args.batch,)).astype(np.int32), device="cuda")   # int32: no dtype-conversion kernel in the step
    gd = np.random.Generator(np.random.PCG64(42 + rank))            # augmentation decisions, host side
    randaug = aug.RandAugment(2, 9, elementwise=args.elementwise)
    autoaug = aug.AutoAugment(elementwise=args.elementwise)
    if args.unfused_augment:
        randaug._transform.fused = False
        autoaug.fused = False

    hw = (args.image_size, args.image_size)

    def autoaugment_decision():
        pol = int(gd.integers(0, 25))
        sub = aug.augmentation_schemes._AUTO_AUGMENT_POLICY_V0[pol]
        return {"policy": pol, "apply": tuple(bool(gd.uniform() < p) for (_t, p, _m) in sub), "negate": (bool(gd.uniform() < 0.5), bool(gd.uniform() < 0.5))}

    reuse = {}

    def step():
        x, plan = images, None
        if not args.no_augment:
            if args.elementwise and os.environ.get("CHB_BENCH_REUSE_PLAN") == "1" and "plan" in reuse:
                plan = reuse["plan"]        # diagnostic: the first step's per-image plan again (no host work, no uploads)
            elif args.elementwise:
                # per-image decisions (elementwise=True, the reference's tf.map_fn mode): every image's own chain, evaluated inside the
                # engine's normalise + patchify pass like the batch-shared chain
                if args.augment == "autoaugment":
                    plan = autoaug.items_plan(images.shape, [autoaugment_decision() for _ in range(args.batch)])
                else:
                    plan = randaug.items_plan(images.shape, [[{"op": int(gd.integers(0, 16)), "negate": bool(gd.uniform() < 0.5),
                                                              "centers": (int(gd.integers(0, hw[0])), int(gd.integers(0, hw[1])))}
                                                             for _ in range(2)] for _ in range(args.batch)])
            elif args.unfused_augment:
                x = (autoaug(images, training=True, decision=autoaugment_decision()) if args.augment == "autoaugment" else
                     randaug(images, training=True, decisions=draw_randaugment_decisions(gd, 2, args.batch, *hw)))
            else:
                # batch-shared decisions (the schemes' default): the chain is evaluated inside the engine's normalise + patchify pass
                plan = (autoaug.plan(images.shape, autoaugment_decision()) if args.augment == "autoaugment" else
                        randaug.plan(images.shape, draw_randaugment_decisions(gd, 2, args.batch, *hw)))
        reuse["plan"] = plan
        return eng.train_step(x, labels, augment=plan, learning_rate=1e-3, weight_decay=0.05)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    red = eng.reducer
    red.measure = True
    red.n_collectives = red.bytes_reduced = 0
    timer.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_in_region = time.perf_counter() - t0         # how long the Python thread took to hand over the K steps (no sync inside):
                                                      # NOT its own cost - once the HIP queue is full every launch call blocks until
                                                      # the GPU has retired one, so this follows the GPU time (host_enqueue below)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.stop()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.mean().item())

    # The host's own share of a step: one step handed over to an EMPTY queue (device idle, nothing to wait for), timed from the first
    # Python statement of the step to the return of its last launch call; median of 5.  Decisions, plans, ctypes transitions and the
    # HIP runtime's launch path, without the back-pressure of a full queue.
    dp_counts = (red.n_collectives, red.bytes_reduced, red.exposed_ms())      # of the timed region only
    red.measure = False
    host_single = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        host_single.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_enqueue = float(np.median(host_single))

    # N > 1: the same K steps again with the gradient exchange switched off (every rank keeps its local gradient): the
    # difference to the timed region above is what the exchange costs a step, waits and CU contention included
    ms_no_exchange = None
    if dist is not None:
        red.measure = False
        red.active = False
        step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_no_exchange = 1e3 * float(t.item()) / args.steps
        red.active = True

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.batch * args.steps / elapsed
        ks = timer.summary()
        fams = KernelTimer.families(ks)
        dom = max(fams, key=lambda k: fams[k]["total_ms"])
        train_flops = 3.0 * forward_flops_per_image(cfg) * args.batch
        out = {
            "metric": "images/sec ViT-B/16 224^2 train step (synthetic)" if args.model == "vitb16" and args.image_size == 224
            else "images/sec %s %d^2 train step (synthetic)" % (args.model, args.image_size),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "host_enqueue_ms_per_step": 1e3 * host_enqueue,
            "host_handover_in_timed_region_ms_per_step": 1e3 * host_in_region / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s train step, batch %d/GPU, %dx%d, on-GPU %s%s, dropout 0.1, AdamW, dp%d"
                       % (args.model, args.batch, args.image_size, args.image_size,
                          "AutoAugment(policy v0)" if args.augment == "autoaugment" else "RandAugment(n=2,m=9)",
                          " OFF" if args.no_augment else (" elementwise" if args.elementwise else (" op-by-op" if args.unfused_augment else " fused")),
                          world),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world},
            # dominant kernel FAMILY (all template instantiations together): achieved = sum of 2*M*N*K over its launches / sum of
            # their durations (HIP events on the launch stream, inside the timed region)
            "roofline": {"bound": "mfma", "kernel": dom + "<*>", "achieved": fams[dom]["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": fams[dom]["tflops"] / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                         "avg_launch_us": fams[dom]["avg_us"], "launches": fams[dom]["launches"],
                         "algorithmic_flop_per_launch": fams[dom]["work"] / fams[dom]["launches"],
                         "note": ("launch durations include time shared with the weight-gradient GEMMs of the engine's second stream "
                                  "(overlap_wgrad); see mfma_all_gemms; the default (CHB_OVERLAP_WGRAD unset) runs one stream")
                         if getattr(eng, "overlap_wgrad", False) else "one stream: launches do not overlap",
                         "families": {k: {"total_ms": round(v["total_ms"], 2), "tflops": round(v["tflops"], 1), "launches": v["launches"],
                                          "frac": round(v["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4)} for k, v in fams.items()},
                         # what a probe that does nothing else measured on this chip (not `peak`: context for `frac`, DESIGN section 4
                         # "Round 3" point 5): the MFMA stream alone on real operand bits, and the K-loop of a 256x256 tile with its
                         # operands streaming in (every placement of the LDS-DMA pieces tried)
                         "measured_ceilings_tflops": {"mfma_stream_only": 2060.0, "kloop_256x256_tile": 1600.0, "kloop_plus_k768_tile_stores": 1300.0,
                                                      "source": "profiles/r03_mfma_power_probe.txt (tools/probe/mfma_ta_probe.hip)"}},
            # all GEMM launches together: their FLOPs over the wall time in which at least one of them ran.  The engine issues a
            # block's weight-gradient GEMMs on a second stream beside the dgrad chain (they fill launch tails, launch gaps and
            # the bandwidth-bound kernels' idle MFMA pipes) when CHB_OVERLAP_WGRAD=1; a launch's own duration - what `roofline` prices, as
            # the contract says - then includes time in which it shared the chip.  Default: one stream, the two numbers agree
            "mfma_all_gemms": {"tflops": sum(v["work"] for v in fams.values()) / (timer.union_ms() * 1e-3) / 1e12,
                               "frac": sum(v["work"] for v in fams.values()) / (timer.union_ms() * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                               "busy_ms_per_step": timer.union_ms() / args.steps,
                               "overlap_wgrad": bool(getattr(eng, "overlap_wgrad", False))},
            "step_tflops": train_flops / (ms_per_step * 1e-3) / 1e12,
            "step_frac_of_mfma_peak": train_flops / (ms_per_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "kernels": {k: {"avg_us": round(v["avg_us"], 2), "tflops": round(v["tflops"], 1), "launches": v["launches"],
                            "total_ms": round(v["total_ms"], 2)} for k, v in ks.items()},
            "final_loss": final_loss,
            # the data-parallel exchange, as this run did it: collectives per step, bytes all-reduced per step per rank, the time
            # per step the compute stream stood waiting for them (HIP events around the reducer's waits on rank 0), and the
            # exposed cost = this step minus the same step re-timed with the exchange off (max over ranks, same K steps)
            "dp": {"backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if dist is not None else None,
                   "payload": args.grad_payload, "forced_single_rank_group": bool(args.force_dp and world == 1),
                   "world_size": dist.get_world_size() if dist is not None else 1,
                   "allreduce_bytes_per_step": dp_counts[1] / args.steps, "collectives_per_step": dp_counts[0] / args.steps,
                   "reducer_wait_ms_per_step": dp_counts[2] / args.steps,
                   "ms_per_step_without_exchange": ms_no_exchange,
                   "exposed_comm_ms_per_step": (ms_per_step - ms_no_exchange) if ms_no_exchange is not None else 0.0,
                   "gradient_bytes": int(eng.G.numel() * eng.G.element_size())},
        }
        # HBM-side bytes per launch of the dominant kernel: PMC counters need their own rocprofv3 passes (tools/pmc_bench.sh runs
        # them over this same command); the committed summary is quoted here when it was taken on this workload
        tr = pmc_traffic(dom) if (args.model == "vitb16" and args.batch == 512 and args.image_size == 224) else None
        if tr is not None:
            out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr["source"]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg_kwargs)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
