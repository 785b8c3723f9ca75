#!/bin/bash
# After `gpurun -- bash tools/final_profiles.sh`: copy what is to be judged from gpurun_out/ (scratch) into profiles/ (tracked), named per round.
set -e
R=r03
cd "$(dirname "$0")/.."
last() { python - "$1" <<'PY'
import sys
print(open(sys.argv[1]).read().strip().splitlines()[-1])
PY
}
last gpurun_out/f_b512.json > profiles/${R}_bench_n1_b512.json
last gpurun_out/f_c4.json > profiles/${R}_bench_n1_config4_vitl16_b512.json
last gpurun_out/f_c5.json > profiles/${R}_bench_n1_config5_vitb16_384_b128_autoaugment.json
last gpurun_out/f_c5_nopad.json > profiles/${R}_bench_n1_config5_unpadded_m.json
last gpurun_out/f_ew.json > profiles/${R}_bench_n1_b512_elementwise.json
last gpurun_out/f_ov.json > profiles/${R}_bench_n1_b512_overlap_wgrad.json
last gpurun_out/f_dp1_fp32.json > profiles/${R}_bench_dp1_forced_rccl_fp32.json
last gpurun_out/f_dp1_bf16.json > profiles/${R}_bench_dp1_forced_rccl_bf16.json
last gpurun_out/f_dp1_queue.json > profiles/${R}_bench_dp1_forced_rccl_tile_queue.json
cp gpurun_out/bench_pmc_traffic.json profiles/${R}_bench_pmc_traffic.json
cp "$(ls -t gpurun_out/prof_b512/*/*_kernel_stats.csv | head -1)" profiles/${R}_bench_b512_kernel_stats.csv
grep -v amdgpu.ids gpurun_out/gemm_bench_final.txt > profiles/${R}_gemm_microbench.txt
grep -v amdgpu.ids gpurun_out/attn_bench_final.txt > profiles/${R}_attention_bench.txt
grep -v amdgpu.ids gpurun_out/augment_stage_final.txt > profiles/${R}_augment_stage.txt
[ -f gpurun_out/r3_fp_bar_depth2.txt ] && grep -v amdgpu.ids gpurun_out/r3_fp_bar_depth2.txt > profiles/${R}_fp_bar_depth2.txt
tail -1 gpurun_out/f_inf.txt > profiles/${R}_inference_latency.txt
ls -la profiles/${R}_*
