#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/pmc_mfma.sh <tag>
# MFMA-pipe utilisation of the GEMM kernels from hardware counters: SQ_VALU_MFMA_BUSY_CYCLES (cycles the MFMA
# pipe of a SIMD is busy, summed over the chip's 1024 SIMDs; = 32 cycles per v_mfma_f32_32x32x16_bf16, 16 per v_mfma_f32_16x16x32_bf16) and
# GRBM_GUI_ACTIVE (active cycles summed over the 8 XCDs).  Counters only with --kernel-trace.
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-parity-mode --no-batcher-leg --no-extra-legs --windows 1 > $R/gpurun_out/pmc_${tag}_mfma.log 2>&1
echo mfma_exit=$?
python - <<PY
import collections, csv, glob, json, sys
sys.path.insert(0, "$R/tools")
from pmc_summary import key_of
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$R/gpurun_out/pmc_${tag}_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[key_of(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, c in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n = c["GRBM_GUI_ACTIVE"][0]
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / max(c["SQ_VALU_MFMA_BUSY_CYCLES"][0], 1)
    act = c["GRBM_GUI_ACTIVE"][1] / max(n, 1)
    if busy < 1e6:
        continue
    # kernel cycles = GRBM_GUI_ACTIVE / 8 (sum over the XCDs); 1024 SIMDs each with one MFMA pipe
    out[k] = {"launches_profiled": n, "mfma_busy_cycles_per_launch": busy, "gui_active_cycles_per_launch": act,
              "mfma_pipe_utilisation": busy / (act / 8.0 * 1024.0)}
json.dump({"counters": "SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE (rocprofv3 --pmc, one pass)",
           "utilisation": "busy cycles summed over 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs * 1024): fraction of the "
                          "kernel's OWN cycles (the chip clocks ~1.7 GHz under this load, not 2.4)",
           "kernels": out}, open("$R/gpurun_out/${tag}_mfma_summary.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"]):
    print("%-48s n=%3d  MFMA pipe utilisation %.3f" % (k, v["launches_profiled"], v["mfma_pipe_utilisation"]))
PY
find $R/gpurun_out/pmc_${tag}_mfma -name '*kernel_trace.csv' -delete
