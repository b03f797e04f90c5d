#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/pmc_step.sh <tag>
# Two separate counter passes over the bench step (TCC counters do not fit one pass); counters
# only with --kernel-trace, never with the sys/hip/hsa trace domains.
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-parity-mode --no-batcher-leg --no-extra-legs --windows 1 > $R/gpurun_out/pmc_${tag}_fetch.log 2>&1
echo fetch_exit=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${tag}_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-parity-mode --no-batcher-leg --no-extra-legs --windows 1 > $R/gpurun_out/pmc_${tag}_write.log 2>&1
echo write_exit=$?
python $R/tools/pmc_summary.py $R/gpurun_out $tag $R/gpurun_out/${tag}_pmc_summary.json
# drop the bulky raw traces from what gets merged back (summary json + logs stay)
find $R/gpurun_out/pmc_${tag}_fetch $R/gpurun_out/pmc_${tag}_write -name '*kernel_trace.csv' -delete
