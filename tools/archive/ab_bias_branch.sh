#!/bin/bash
# same-box A/B: the store epilogue without the add of a null bias (PCAA_V2_BIAS_BRANCH, compile-time); run on the GPU box
set -e
Q="--no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs"
python -m pytest tests/test_hip_ops.py tests/test_hip_modules.py -m gpu -x -q 2>&1 | tail -2
echo "== bias branch (default build)"; python tools/gemm_lab.py --rounds 3 --variants 0:0 | grep -E "fwd"
for n in 128 32; do python bench.py $Q --points $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('branch N=$n step', round(d['ms_per_step'],3), d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'))"; done
echo "== folded add"; PCAA_HIPCC_EXTRA=-DPCAA_V2_BIAS_BRANCH=0 python -m opensetgaitrecognition_pcaa_amd.build > /dev/null
export PCAA_HIPCC_EXTRA=-DPCAA_V2_BIAS_BRANCH=0
python tools/gemm_lab.py --rounds 3 --variants 0:0 | grep -E "fwd"
for n in 128 32; do python bench.py $Q --points $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('folded N=$n step', round(d['ms_per_step'],3), d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'))"; done
