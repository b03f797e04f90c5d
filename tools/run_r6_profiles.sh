#!/bin/bash
# on the GPU box (via gpurun): round-6 profile set of the default bench workload (config[1], bf16)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_step.sh r06 && python tools/step_breakdown.py gpurun_out/prof_r06 48 > gpurun_out/r06_step_breakdown.txt; head -12 gpurun_out/r06_step_breakdown.txt
find gpurun_out/prof_r06 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_bf16_kernel_stats.csv
# the same trace reduced to its 5 timed steps (rocprofv3's own summary averages the 2 warm-up steps in)
python tools/steady_kernel_stats.py gpurun_out/prof_r06 --steps 5 --flop-per-launch 257.7e9 > gpurun_out/r06_steady_kernel_stats.csv 2> gpurun_out/r06_steady_kernel_stats.txt; cat gpurun_out/r06_steady_kernel_stats.txt
python tools/trace_list.py gpurun_out/prof_r06 > gpurun_out/r06_trace_list.txt 2>&1
python tools/step_sections.py > gpurun_out/r06_step_sections.txt 2>&1; tail -14 gpurun_out/r06_step_sections.txt
bash tools/pmc_step.sh r06 | tail -12
bash tools/pmc_mfma.sh r06 | tail -8
find gpurun_out/prof_r06 -name "*.csv" -size +2M -delete
# round 6: one rank's program of an emulated 8-rank job (packed gathered operands), same trace -> where its extra 0.4 ms goes
bash tools/prof_step.sh r06_emu8 --dp-emulate 8 && python tools/step_breakdown.py gpurun_out/prof_r06_emu8 48 > gpurun_out/r06_emu8_step_breakdown.txt; head -14 gpurun_out/r06_emu8_step_breakdown.txt
python tools/trace_list.py gpurun_out/prof_r06_emu8 > gpurun_out/r06_emu8_trace_list.txt 2>&1
find gpurun_out/prof_r06_emu8 -name "*.csv" -size +2M -delete
