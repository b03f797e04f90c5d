#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a list: start offset, duration, queue, kernel -- to spot small kernels that
take long on the critical path.   python tools/trace_list.py <rocprof output dir> [min_us]"""
import csv, glob, sys
fs = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=lambda f: -len(open(f).readlines()))
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r['Start_Timestamp']))
ce = [i for i, r in enumerate(rows) if 'cross_entropy_kernel' in r['Kernel_Name']]
seg = rows[ce[-2]: ce[-1]]
t0 = int(seg[0]['Start_Timestamp'])
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d < mn:
        continue
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} +{d:7.1f} us  q{r['Queue_Id']}  grid {r['Grid_Size_X']:>8}  {name[:70]}")
