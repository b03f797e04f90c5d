#!/bin/bash
# usage (GPU box): bash tools/l2_exp.sh  -- timing sweep of PCAA_GEMM_STAGGER, then L2 PMC passes
R=$GRAFT_REPO_ROOT
for st in 0 1 2 4 8; do
  for sh in "512 512" "512 1024" "1024 1024"; do
    set -- $sh
    PCAA_GEMM_STAGGER=$st python $R/tools/gemm_l2.py --cin $1 --cout $2
  done
done
cd /tmp && export TMPDIR=/tmp
for st in 0 2; do
  export PCAA_GEMM_STAGGER=$st
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/l2_hit_$st -- python $R/tools/gemm_l2.py --cin 512 --cout 1024 --iters 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/l2_fetch_$st -- python $R/tools/gemm_l2.py --cin 512 --cout 1024 --iters 3 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for d in sorted(glob.glob(R + "/gpurun_out/l2_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_dma" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        print(os.path.basename(d), {k: (v[0], v[1] / max(v[0], 1)) for k, v in agg.items()})
PY
