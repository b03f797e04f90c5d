#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/pmc_gemm.sh <tag>
# Where the LDS-DMA GEMM's wave cycles go (hardware counters, separate passes; counters only with --kernel-trace):
# pass 1: SQ wave-state split; pass 2: LDS; pass 3: L2 hit/miss.
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/${tag}_counters_avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcg_${tag}_$i -- python3 $R/tools/gemm_one.py > $R/gpurun_out/pmcg_${tag}_$i.log 2>&1
  echo "pass $i exit=$?"
done
python3 - <<PY
import collections, csv, glob, json, sys
sys.path.insert(0, "$R/tools")
from pmc_summary import key_of
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$R/gpurun_out/pmcg_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key_of(r["Kernel_Name"])
        if not k.startswith(("gemm_bf16_dma", "gemm_bf16_v2")):
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / max(v[0], 1) for c, v in cs.items()} for k, cs in agg.items()}
json.dump({"what": "per-launch averages on [245760,1024]x[1024,1024] (tools/gemm_one.py), rocprofv3 --pmc, one counter set per pass",
           "kernels": out}, open("$R/gpurun_out/${tag}_gemm_counters.json", "w"), indent=1)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %.4g" % (c, v))
    w = cs.get("SQ_WAVE_CYCLES")
    if w:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in cs: print("   %-28s %.3f of wave cycles" % (c + " /", cs[c] / w))
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
        print("   L2 hit rate %.3f" % (cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])))
PY
find $R/gpurun_out/pmcg_${tag}_* -name '*kernel_trace.csv' -delete
