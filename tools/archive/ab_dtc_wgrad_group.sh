#!/bin/bash
# same-box A/B: the temporal block's six weight gradients as one grouped launch (PCAA_DTC_WGRAD_GROUP); run on the GPU box
python -m pytest tests/test_hip_ops.py tests/test_hip_modules.py tests/test_end_to_end.py -m gpu -x -q 2>&1 | tail -2
Q="--no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs --no-kernel-timing"
for rep in 1 2; do for g in 0 wg main; do for n in 32 64 128; do
PCAA_DTC_WGRAD_GROUP=$g python bench.py $Q --points $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group=$g N=$n', round(d['ms_per_step'],3))"
done; done; done
