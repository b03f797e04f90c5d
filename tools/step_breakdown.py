#!/usr/bin/env python3
"""Aggregate one training step of a rocprofv3 kernel trace by kernel (python tools/step_breakdown.py <dir>)."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# one full period of the step, delimited by a kernel that runs exactly once per step
ad = [i for i, r in enumerate(rows) if 'cross_entropy_kernel' in r['Kernel_Name']]
seg = rows[ad[-2] + 1: ad[-1] + 1]
agg = collections.OrderedDict()
for r in seg:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '').replace('void ', '')
    n = re.sub(r'\(.*', '', n)[:56]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(len(seg), 'kernels; busy %.3f ms (sum of durations, concurrent streams overlap); span %.3f ms' % (
    sum(v[1] for v in agg.values()) / 1e3, (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{k:56s} n={n:3d} {t / 1e3:7.3f} ms")
