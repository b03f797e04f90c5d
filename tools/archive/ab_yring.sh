#!/bin/bash
# same-box A/B of the fused-dgrad epilogue's y prefetch depth (PCAA_V2_YRING, a compile-time constant): run on the GPU box
set -e
Q="--no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs"
python -m pytest tests/test_hip_ops.py -m gpu -x -q 2>&1 | tail -2
echo "== ring 4 (default build)"; python tools/gemm_lab.py --rounds 3 --variants 1:0 | grep -E "dgrad_bn|fwd " 
python bench.py $Q | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_timed'])"
PCAA_GEMM_V2_RC=0 python bench.py $Q | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step RC=0', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_timed'])"
echo "== ring 2 (rounds 1-3)"; PCAA_HIPCC_EXTRA=-DPCAA_V2_YRING=2 python -m opensetgaitrecognition_pcaa_amd.build > /dev/null
PCAA_HIPCC_EXTRA=-DPCAA_V2_YRING=2 python tools/gemm_lab.py --rounds 3 --variants 1:0 | grep -E "dgrad_bn|fwd "
PCAA_HIPCC_EXTRA=-DPCAA_V2_YRING=2 python bench.py $Q | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_timed'])"
