#!/bin/bash
# on the one-GPU box (via gpurun): rehearse bench.py's N>1 branches -- 2 ranks sharing the GPU over gloo (default
# exchange, ZeRO-1, bf16 gradient compression), and RCCL on a 1-rank group (--dp-force)
R=$GRAFT_REPO_ROOT
cd $R
for extra in "" "--dp-mode zero" "--grad-compress bf16" "--sync-bn"; do
  PCAA_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --no-kernel-timing --no-cpu-baseline --no-extra-legs --windows 1 $extra 2> gpurun_out/dp_reh.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gloo x2 [$extra]', round(d['ms_per_step'],2), 'ms/step', d['n_gpus'], 'ranks', d['config']['dp'])"
done
python bench.py --dp-force --steps 10 --warmup 3 --no-kernel-timing --no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs --windows 1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rccl x1 forced', round(d['ms_per_step'],2), 'ms/step', d['config']['dp'])"
python bench.py --steps 10 --warmup 3 --no-kernel-timing --no-cpu-baseline --no-parity-mode --no-batcher-leg --no-extra-legs --windows 1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no collectives', round(d['ms_per_step'],2), 'ms/step')"
tail -3 gpurun_out/dp_reh.err
