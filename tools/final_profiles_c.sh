#!/bin/bash
# Short refresh of the round's headline files on the final build (tools/final_profiles.sh is the full recipe): -> gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b512
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b512 -- python bench.py --no-cpu-baseline > gpurun_out/prof_b512.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/f_b512.json 2> gpurun_out/f_b512.err
timeout -k 10 300 python bench.py --model vitl16 --batch 512 --no-cpu-baseline > gpurun_out/f_c4.json 2>/dev/null
timeout -k 10 300 python bench.py --image-size 384 --batch 128 --augment autoaugment --no-cpu-baseline > gpurun_out/f_c5.json 2>/dev/null
timeout -k 10 200 python tools/attn_bench.py 128 577 12 > gpurun_out/attn_bench_577.txt 2>&1
for f in f_b512 f_c4 f_c5; do
  python -c "import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), round(d['value'],1), 'host', round(d.get('host_enqueue_ms_per_step',0),2), d['roofline']['kernel'], round(d['roofline']['frac'],4))"
done
find gpurun_out/prof_b512 -name "*kernel_stats.csv" | head -1
