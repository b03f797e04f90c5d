#!/bin/bash
# usage (GPU box): bash tools/wgrad_exp.sh -- wgrad tile->XCD mapping experiment (time + fabric fetch bytes)
R=$GRAFT_REPO_ROOT
for sf in 0 1; do
  for sh in "512 512 64" "512 1024 32" "1024 1024 16"; do
    set -- $sh
    if [ $sf = 1 ]; then export PCAA_GEMM_SPLIT_FAST=1; else unset PCAA_GEMM_SPLIT_FAST; fi
    python $R/tools/gemm_l2.py --mode wgrad --cin $1 --cout $2 --split $3 | sed "s/^/split_fast=$sf /"
  done
done
cd /tmp && export TMPDIR=/tmp
for sf in 0 1; do
  if [ $sf = 1 ]; then export PCAA_GEMM_SPLIT_FAST=1; else unset PCAA_GEMM_SPLIT_FAST; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/wg_fetch_$sf -- python $R/tools/gemm_l2.py --mode wgrad --cin 1024 --cout 1024 --split 16 --iters 3 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for d in sorted(glob.glob(R + "/gpurun_out/wg_fetch_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_dma" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        print(os.path.basename(d), {k: (v[0], 2 * 1024 * v[1] / max(v[0], 1) / 1e6) for k, v in agg.items()}, "MB/launch (corrected)")
PY
