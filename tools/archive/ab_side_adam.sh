#!/bin/bash
# same-box A/B: where the decoder's side-stream update starts (PCAA_SIDE_ADAM_AT); run on the GPU box
Q="--no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs --no-kernel-timing"
for at in dec_bwd heads pointnet dec_bwd pointnet; do
  for n in 128 32; do
    PCAA_SIDE_ADAM_AT=$at python bench.py $Q --points $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$at N=$n', round(d['ms_per_step'],3), d.get('windows_ms_per_step'))"
  done
done
