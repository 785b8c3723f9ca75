#!/bin/bash
# Second session of the round's profile recipe (micro-benchmarks; tools/final_profiles.sh is the first): -> gpurun_out/ (copy to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/gemm_bench.py 20 > gpurun_out/gemm_bench_final.txt 2>&1
timeout -k 10 200 python tools/attn_bench.py > gpurun_out/attn_bench_final.txt 2>&1
timeout -k 10 300 python tools/attn_pipe_check.py --time-only --barrier-experiment > gpurun_out/attn_pipe_ablation_final.txt 2>&1
timeout -k 10 600 python tools/rccl_contention.py 512 32 300 > gpurun_out/contention_final.txt 2>&1
timeout -k 10 300 python tools/augment_stage_bench.py > gpurun_out/augment_stage_final.txt 2>&1
timeout -k 10 200 python tools/inference_latency.py 256 > gpurun_out/f_inf.txt 2>&1
tail -1 gpurun_out/f_inf.txt
timeout -k 10 600 bash tools/pmc_attn.sh > gpurun_out/attn_pmc_final.txt 2>&1
tail -30 gpurun_out/attn_bench_final.txt
