#!/usr/bin/env python3
"""How many event records / waits one train step puts on each stream (round 6: an event record on the main stream between two
kernels costs ~7 us of idle queue -- rocprofv3 trace: the gaps behind every _on_wgrad_stream entry).
    python tools/lab/event_count.py [--dp-emulate 8]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from opensetgaitrecognition_pcaa_amd import constants, synthetic as syn
from opensetgaitrecognition_pcaa_amd.train import PCAATrainer
from opensetgaitrecognition_pcaa_amd.utils import sample_distant_points

ap = argparse.ArgumentParser()
ap.add_argument("--dp-emulate", type=int, default=0)
a = ap.parse_args()
B, N, C, K, T = 64, 128, 4, 8, constants.NSTEPS
constants.NFEATURES = C
cfg = dict(constants.CONFIG); cfg.update(NMAX=N, TRAIN_CLASSES=list(range(K)), BATCH_SIZE=B)
tr = PCAATrainer(cfg, precision="bf16", emulate_world=a.dp_emulate, dp_gather=bool(a.dp_emulate), grad_compress="bf16" if a.dp_emulate else None)
tr.set_prior_means(sample_distant_points(32, K, 10, 10)); tr.finalize(); tr.train()
inp = (syn.synthetic_pcs(B, T, N, C).cuda().permute(0, 3, 1, 2), syn.synthetic_labels(B, K).cuda(), syn.synthetic_z0(B, 32).cuda(),
       syn.synthetic_alphas(B).cuda())
for _ in range(3):
    tr.step(*inp)
torch.cuda.synchronize()
names = {torch.cuda.current_stream().cuda_stream: "main", tr._side.cuda_stream: "side(adam)", tr._aux.cuda_stream: "aux(critic)",
         tr._wg.cuda_stream: "wgrad"}
rec, waits = collections.Counter(), collections.Counter()
orig_record, orig_wait_event, orig_wait_stream = torch.cuda.Event.record, torch.cuda.Stream.wait_event, torch.cuda.Stream.wait_stream


def record(self, stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    rec[names.get(s.cuda_stream, hex(s.cuda_stream))] += 1
    return orig_record(self, stream) if stream is not None else orig_record(self)


def wait_event(self, ev):
    waits[names.get(self.cuda_stream, hex(self.cuda_stream))] += 1
    return orig_wait_event(self, ev)


def wait_stream(self, other):
    waits[names.get(self.cuda_stream, hex(self.cuda_stream))] += 1
    rec[names.get(other.cuda_stream, hex(other.cuda_stream)) + " (via wait_stream)"] += 1
    return orig_wait_stream(self, other)


torch.cuda.Event.record, torch.cuda.Stream.wait_event, torch.cuda.Stream.wait_stream = record, wait_event, wait_stream
tr.step(*inp)
torch.cuda.synchronize()
print("event records per step, by stream:", dict(rec))
print("event waits per step, by waiting stream:", dict(waits))
