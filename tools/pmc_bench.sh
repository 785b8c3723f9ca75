#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters: two separate passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only —
# gpurun refuses --pmc together with sys/hip traces) over a short run of the same bench.py command, then per-kernel
# per-launch averages with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE (KB) x 2 for
# wide coalesced reads, WRITE_SIZE (KB) as is.  Output: gpurun_out/bench_pmc_traffic.json (copy to profiles/).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcb_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcb_$c.log 2>&1 || exit 1
done
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcb_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(agg.items()):
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    if not f or not w:
        continue
    fetch = 2.0 * 1024.0 * sum(f) / len(f)
    write = 1024.0 * sum(w) / len(w)
    out[k] = {"launches_sampled": len(f), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": fetch + write, "raw_FETCH_SIZE_KB": sum(f) / len(f), "raw_WRITE_SIZE_KB": sum(w) / len(w)}
json.dump({"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "corrections": "FETCH_SIZE KB x2 (gfx950 wide-read under-count), WRITE_SIZE KB exact; averages over all launches of a kernel name",
           "kernels": out}, open("gpurun_out/bench_pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    print("%-60s n=%4d  fetch %10.1f MB  write %10.1f MB" % (k[:60], v["launches_sampled"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
PY
