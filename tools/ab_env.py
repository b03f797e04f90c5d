#!/usr/bin/env python3
"""Same-box A/B of environment switches: python tools/ab_env.py VAR=a,b,c [--reps 3] [--steps 30] [-- bench args]
Runs bench.py (no CPU baseline, no kernel timing) reps times per value, interleaved, prints the median ms/step."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
reps = int(args[args.index("--reps") + 1]) if "--reps" in args else 3
steps = args[args.index("--steps") + 1] if "--steps" in args else "30"
var, vals = [a for a in args if "=" in a][0].split("=", 1)
vals = vals.split(",")
res = {v: [] for v in vals}
for _ in range(reps):
    for v in vals:
        env = dict(os.environ); env[var] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-kernel-timing",
                              "--steps", steps] + extra, capture_output=True, text=True, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if line:
            res[v].append(json.loads(line[-1])["ms_per_step"])
for v in vals:
    r = res[v]
    print(f"{var}={v:10s} median {statistics.median(r):7.3f} ms/step   runs {[round(x, 3) for x in r]}", flush=True)
