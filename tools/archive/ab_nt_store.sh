#!/bin/bash
# same-box A/B of the non-temporal C stores of the 4-wave GEMM loop (PCAA_V2_NT_STORE, compile-time); run on the GPU box
set -e
Q="--no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs"
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
echo "== nt stores (default build)"; python tools/gemm_lab.py --rounds 3 --variants 0:0 | grep -E "fwd|dgrad_bn"
for n in 128 32; do python bench.py $Q --points $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt N=$n step', round(d['ms_per_step'],3), d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'))"; done
echo "== plain stores"; PCAA_HIPCC_EXTRA=-DPCAA_V2_NT_STORE=0 python -m opensetgaitrecognition_pcaa_amd.build > /dev/null
export PCAA_HIPCC_EXTRA=-DPCAA_V2_NT_STORE=0
python tools/gemm_lab.py --rounds 3 --variants 0:0 | grep -E "fwd|dgrad_bn"
for n in 128 32; do python bench.py $Q --points $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain N=$n step', round(d['ms_per_step'],3), d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'))"; done
