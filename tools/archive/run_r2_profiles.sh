#!/bin/bash
# on the GPU box (via gpurun): round-2 profile set of the default bench workload
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_step.sh r02 && python tools/step_breakdown.py gpurun_out/prof_r02 40 > gpurun_out/r02_step_breakdown.txt; cat gpurun_out/r02_step_breakdown.txt | head -30
find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_bf16_kernel_stats.csv
python tools/step_sections.py > gpurun_out/r02_step_sections.txt 2>&1; tail -16 gpurun_out/r02_step_sections.txt
bash tools/pmc_step.sh r02 | tail -25
bash tools/pmc_mfma.sh r02 | tail -8
find gpurun_out/prof_r02 -name "*.csv" -size +2M -delete
