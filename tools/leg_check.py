#!/usr/bin/env python3
"""Does the step time depend on how many trainers the process has built before?  (Round 3: it did -- each trainer got
fresh side streams, HIP maps streams onto four hardware queues round-robin, and from the fourth trainer on a side stream
shared the main stream's queue: 6.0 instead of 5.7 ms/step.  train._side_streams now hands every trainer the same
streams.)  Builds four trainers one after another, deterministic and device-side weight fills alternating, and times
5 windows of 20 steps each:  python tools/leg_check.py"""
import sys, json, statistics, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, bench
from opensetgaitrecognition_pcaa_amd import constants, functional as F_hip
sys.argv = ['bench.py']
a = bench.parse()
dev = torch.device('cuda', 0)
constants.NFEATURES = 4
F_hip.set_precision('bf16')
for trial in range(2):
    for fill in ('deterministic', 'device'):
        tr, _ = bench.build_trainer(a, 128, dev, None, 'bf16', fill=fill)
        inp = bench.make_inputs(64, 30, 128, 4, 8, dev)
        for _ in range(5): tr.step(*inp)
        ms, _ = bench.time_single_gpu(lambda: tr.step(*inp), 20, 5)
        print(trial, fill, [round(m, 3) for m in ms], flush=True)
        del tr; torch.cuda.empty_cache()
