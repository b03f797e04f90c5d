#!/bin/bash
# same-box A/B of the two-sequence temporal kernels in the step, per direction (run on the GPU box)
Q="--no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs --no-kernel-timing"
for rep in 1 2; do
  for cfg in "PCAA_DTC_PAIR=0" "PCAA_DTC_PAIR=fwd" "PCAA_DTC_PAIR=adj" "PCAA_DTC_PAIR=1"; do
    for n in 128 32; do
      env $cfg python bench.py $Q --points $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg N=$n', round(d['ms_per_step'],3))"
    done
  done
done
