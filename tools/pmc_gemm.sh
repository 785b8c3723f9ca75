#!/bin/bash
# PMC passes over the GEMM micro-benchmark (separate passes per counter group, kernel-trace only: gpurun refuses --pmc with sys/hip traces)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_$tag -- python tools/gemm_bench.py 3 ${1:-all} > gpurun_out/pmc_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        k = k.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        key = (k, r["Grid_Size"] if "Grid_Size" in r else "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
