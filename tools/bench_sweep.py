#!/usr/bin/env python3
"""BASELINE config[3]: point-subsampling sweep N in {32, 64, 128, 256} (reference train_pointsubsampling.py drives
train_variant4 with these NMAX values) -- one bench.py run per N, B=64, bf16 mode: python tools/bench_sweep.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for n in (int(a) for a in (sys.argv[1:] or ["32", "64", "128", "150", "256"])):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-kernel-timing",
                          "--points", str(n), "--steps", "10"], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"N={n}: failed\n{out.stderr[-400:]}")
        continue
    d = json.loads(line[-1])
    print(f"N={n:4d}  {d['ms_per_step']:7.3f} ms/step  {d['value']:8.0f} sequences/s", flush=True)
