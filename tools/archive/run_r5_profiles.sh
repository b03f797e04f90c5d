#!/bin/bash
# on the GPU box (via gpurun): round-5 profile set of the default bench workload (config[1], bf16)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_step.sh r05 && python tools/step_breakdown.py gpurun_out/prof_r05 48 > gpurun_out/r05_step_breakdown.txt; head -12 gpurun_out/r05_step_breakdown.txt
find gpurun_out/prof_r05 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_bf16_kernel_stats.csv
# the same trace reduced to its 5 timed steps (rocprofv3's own summary averages the 2 warm-up steps in)
python tools/steady_kernel_stats.py gpurun_out/prof_r05 --steps 5 --flop-per-launch 257.7e9 > gpurun_out/r05_steady_kernel_stats.csv 2> gpurun_out/r05_steady_kernel_stats.txt; cat gpurun_out/r05_steady_kernel_stats.txt
python tools/trace_list.py gpurun_out/prof_r05 > gpurun_out/r05_trace_list.txt 2>&1
python tools/step_sections.py > gpurun_out/r05_step_sections.txt 2>&1; tail -14 gpurun_out/r05_step_sections.txt
bash tools/pmc_step.sh r05 | tail -12
bash tools/pmc_mfma.sh r05 | tail -8
find gpurun_out/prof_r05 -name "*.csv" -size +2M -delete
