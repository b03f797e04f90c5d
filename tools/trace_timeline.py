#!/usr/bin/env python3
"""Timeline of ONE train step from a rocprofv3 kernel trace (CSV): span, union busy time (some kernel running),
idle gaps by size, concurrency, and the per-stream launch counts -- to compare eager enqueue with hipGraph replay.

    python tools/trace_timeline.py <rocprof output dir> [label]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ce = [i for i, r in enumerate(rows) if 'cross_entropy_kernel' in r['Kernel_Name']]
seg = rows[ce[-2]: ce[-1]]                 # one full period of the step
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
span = (max(e for _, e in iv) - iv[0][0]) / 1e3
busy, gaps, cur_s, cur_e = 0.0, [], iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e) / 1e3)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in iv) / 1e3
queues = {}
for r in seg:
    queues[r.get('Queue_Id', '?')] = queues.get(r.get('Queue_Id', '?'), 0) + 1
print(f"[{label}] {len(seg)} kernels, span {span:.0f} us, some kernel running {busy / 1e3:.0f} us, idle {sum(gaps):.0f} us "
      f"in {len(gaps)} gaps (>=5us: {sum(1 for g in gaps if g >= 5)} = {sum(g for g in gaps if g >= 5):.0f} us; "
      f">=20us: {sum(1 for g in gaps if g >= 20)} = {sum(g for g in gaps if g >= 20):.0f} us), "
      f"sum of durations {tot:.0f} us (overlap {tot - busy / 1e3:.0f} us), queues {queues}")
# the biggest gaps and what follows them
order = sorted(seg, key=lambda r: int(r['Start_Timestamp']))
ends = 0
big = []
for r in order:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if ends and s - ends > 8000:
        big.append(((s - ends) / 1e3, r['Kernel_Name'][:70]))
    ends = max(ends, e)
for g, n in sorted(big, reverse=True)[:12]:
    print(f"   gap {g:7.1f} us before {n}")
