#!/bin/bash
# on the GPU box (via gpurun): round-4 profile set of the default bench workload (config[1], bf16)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_step.sh r04 && python tools/step_breakdown.py gpurun_out/prof_r04 48 > gpurun_out/r04_step_breakdown.txt; head -30 gpurun_out/r04_step_breakdown.txt
find gpurun_out/prof_r04 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_bf16_kernel_stats.csv
python tools/trace_list.py gpurun_out/prof_r04 > gpurun_out/r04_trace_list.txt 2>&1
python tools/step_sections.py > gpurun_out/r04_step_sections.txt 2>&1; tail -16 gpurun_out/r04_step_sections.txt
bash tools/pmc_step.sh r04 | tail -25
bash tools/pmc_mfma.sh r04 | tail -8
bash tools/pmc_gemm.sh r04 | tail -40
find gpurun_out/prof_r04 -name "*.csv" -size +2M -delete
