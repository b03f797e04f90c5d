#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_step.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-parity-mode --no-batcher-leg --no-extra-legs --windows 1 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
echo prof_exit=$?
