#!/bin/bash
# One session on the GPU box: PMC traffic, rocprof kernel stats, the bench line and its companions -> gpurun_out/ (copy to profiles/).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
cp gpurun_out/bench_pmc_traffic.json profiles/r02_bench_pmc_traffic.json
rm -rf gpurun_out/prof_b512
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b512 -- python bench.py --no-cpu-baseline > gpurun_out/prof_b512.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/f_b512.json 2> gpurun_out/f_b512.err
timeout -k 10 300 python bench.py --model vitl16 --batch 512 --no-cpu-baseline > gpurun_out/f_c4.json 2>/dev/null
timeout -k 10 300 python bench.py --image-size 384 --batch 128 --augment autoaugment --no-cpu-baseline > gpurun_out/f_c5.json 2>/dev/null
timeout -k 10 300 python bench.py --elementwise --no-cpu-baseline > gpurun_out/f_ew.json 2>/dev/null
CHB_OVERLAP_WGRAD=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/f_ov.json 2>/dev/null
CHB_GEMM_ALGO=2 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/f_algo2.json 2>/dev/null
timeout -k 10 200 python tools/gemm_bench.py 20 > gpurun_out/gemm_bench_final.txt 2>&1
timeout -k 10 200 python tools/inference_latency.py 256 > gpurun_out/f_inf.txt 2>&1
for f in f_b512 f_c4 f_c5 f_ew f_ov f_algo2; do
  python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], {k:round(v['tflops']) for k,v in d['roofline']['families'].items() if '256' in k})"
done
tail -1 gpurun_out/f_inf.txt
