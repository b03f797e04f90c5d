#!/bin/bash
# on the GPU box (via gpurun): the whole -m gpu suite, then the three bench workloads
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -rP > $R/gpurun_out/r2_suite.log 2>&1; echo "pytest rc=$?" >> $R/gpurun_out/r2_suite.log
grep -E "passed|failed|rc=|agreement|rel-l2" $R/gpurun_out/r2_suite.log | tail -12
python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r2_bench_train.json 2> $R/gpurun_out/r2_bench_train.err; tail -c 1500 $R/gpurun_out/r2_bench_train.json
python bench.py --workload sweep --steps 20 --warmup 5 > $R/gpurun_out/r2_bench_sweep.json 2> $R/gpurun_out/r2_bench_sweep.err; tail -c 2500 $R/gpurun_out/r2_bench_sweep.json
python bench.py --workload infer --steps 10 --warmup 3 > $R/gpurun_out/r2_bench_infer.json 2> $R/gpurun_out/r2_bench_infer.err; tail -c 1500 $R/gpurun_out/r2_bench_infer.json
for f in $R/gpurun_out/r2_bench_*.err; do tail -n 3 "$f"; done
