#!/bin/bash
# Per-shape HBM-side traffic of the GEMM micro-benchmark: FETCH_SIZE / WRITE_SIZE passes, dispatches mapped to shapes by order.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcg_FETCH_SIZE gpurun_out/pmcg_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcg_$c -- python tools/gemm_bench.py 2 ${1:-all} > gpurun_out/pmcg_$c.log 2>&1 || exit 1
done
python - <<'PY'
import csv, glob
names = [l.split()[0] for l in open("gpurun_out/pmcg_FETCH_SIZE.log") if " TF/s" in l]
lines = [l.strip() for l in open("gpurun_out/pmcg_FETCH_SIZE.log") if " TF/s" in l]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob("gpurun_out/pmcg_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "gemm_" in r["Kernel_Name"] and r["Counter_Name"] == c:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals[c] = [v for _, v in rows]
per = 3
for i, nm in enumerate(names):
    f = vals["FETCH_SIZE"][i * per:(i + 1) * per]
    w = vals["WRITE_SIZE"][i * per:(i + 1) * per]
    if not f: break
    print("%-60s fetch %8.1f MB (x2 corrected)  write %8.1f MB" % (lines[i][:60], 2 * 1024 * sum(f) / len(f) / 1e6, 1024 * sum(w) / len(w) / 1e6))
PY
