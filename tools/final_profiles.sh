#!/bin/bash
# One session on the GPU box: PMC traffic, rocprof kernel stats, the bench line and its companions -> gpurun_out/ (copy to profiles/).
# Round 4: profiles/r04_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
cp gpurun_out/bench_pmc_traffic.json profiles/r04_bench_pmc_traffic.json
rm -rf gpurun_out/prof_b512
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b512 -- python bench.py --no-cpu-baseline > gpurun_out/prof_b512.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/f_b512.json 2> gpurun_out/f_b512.err
timeout -k 10 300 python bench.py --model vitl16 --batch 512 --no-cpu-baseline > gpurun_out/f_c4.json 2>/dev/null
timeout -k 10 300 python bench.py --image-size 384 --batch 128 --augment autoaugment --no-cpu-baseline > gpurun_out/f_c5.json 2>/dev/null
CHB_PAD_M=0 timeout -k 10 300 python bench.py --image-size 384 --batch 128 --augment autoaugment --no-cpu-baseline > gpurun_out/f_c5_nopad.json 2>/dev/null
timeout -k 10 300 python bench.py --elementwise --no-cpu-baseline > gpurun_out/f_ew.json 2>/dev/null
CHB_OVERLAP_WGRAD=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/f_ov.json 2>/dev/null
CHB_ENGINE_PY_BLOCKS=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/f_pyblocks.json 2>/dev/null
CHB_ATTN_BWD_ALGO=4 CHB_ATTN_FWD_ALGO=3 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/f_oldattn.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --force-dp > gpurun_out/f_dp1_fp32.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --force-dp --grad-payload bf16 > gpurun_out/f_dp1_bf16.json 2>/dev/null
CHB_GEMM_TILE_QUEUE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --force-dp > gpurun_out/f_dp1_queue.json 2>/dev/null
for f in f_b512 f_c4 f_c5 f_c5_nopad f_ew f_ov f_pyblocks f_oldattn f_dp1_fp32 f_dp1_bf16 f_dp1_queue; do
  python -c "import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), round(d['value'],1), 'host', round(d.get('host_enqueue_ms_per_step',0),2), d['roofline']['kernel'], round(d['roofline']['frac'],4), {k:round(v['tflops']) for k,v in d['roofline']['families'].items() if '256' in k}, {k: d['dp'][k] for k in ('backend','payload','collectives_per_step','reducer_wait_ms_per_step','exposed_comm_ms_per_step')})"
done
