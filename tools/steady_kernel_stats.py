#!/usr/bin/env python3
"""Steady-state kernel summary of a rocprofv3 kernel trace of bench.py (evidence hygiene).

rocprofv3's own ``*_kernel_stats.csv`` averages the warm-up launches with the timed ones (round 4: 0.385 against the
bench line's 0.401 on the same trace).  This reads the ``*_kernel_trace.csv`` of the same run, keeps only the LAST
``--steps`` train steps (a step = everything between two ``cross_entropy_kernel`` launches, which run exactly once per
step) and writes the summary in rocprofv3's stats layout, so that the dominant kernel's AverageNs here is the figure
``roofline.achieved`` is computed from.  (The window is rotated by the part of a step that precedes its cross-entropy
launch: it holds the encoder forward of every timed step and the remainder of the step BEFORE each -- the last warm-up
step's backward stands in for the last timed step's; with >= 2 warm-up steps both are steady state.)

    python tools/steady_kernel_stats.py <rocprofv3 output dir> --steps 5 [--flop-per-launch 257.7e9] > profiles/rNN_steady_kernel_stats.csv
"""
import argparse
import csv
import glob
import statistics
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=5, help="timed steps of the traced bench.py run (its --steps)")
    ap.add_argument("--delimiter", default="cross_entropy_kernel")
    ap.add_argument("--dominant", default="gemm_bf16_v2_kernel", help="substring of the kernel family the roofline reports")
    ap.add_argument("--flop-per-launch", type=float, default=None,
                    help="algorithmic FLOP per launch of the dominant kernel: prints achieved TFLOP/s to stderr")
    a = ap.parse_args()
    f = glob.glob(a.dir + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.delimiter in r["Kernel_Name"]]
    if len(marks) < a.steps + 1:
        sys.exit(f"only {len(marks)} step delimiters in the trace, need {a.steps + 1}")
    # the encoder forward of a step precedes its cross-entropy launch: a step's kernels run from just after the
    # PREVIOUS delimiter to its own; the last `steps` whole periods
    seg = rows[marks[-a.steps - 1] + 1: marks[-1] + 1]
    agg = {}
    for r in seg:
        agg.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(d), sum(d), round(sum(d) / len(d), 6), round(100.0 * sum(d) / total, 2), min(d), max(d),
                    round(statistics.pstdev(d), 6)])
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6 / a.steps
    dom = [(n, d) for n, d in agg.items() if a.dominant in n]
    if dom:
        name, d = max(dom, key=lambda kv: sum(kv[1]))
        msg = (f"steady state: last {a.steps} steps, {len(seg)} launches, span {span:.3f} ms/step; dominant {name[:70]}: "
               f"{len(d)} launches, avg {sum(d) / len(d) / 1e3:.2f} us")
        if a.flop_per_launch:
            tf = a.flop_per_launch / (sum(d) / len(d) * 1e-9) / 1e12
            msg += f" -> {tf:.1f} TFLOP/s = {tf / 2500.0:.3f} of 2 500"
        print(msg, file=sys.stderr)


if __name__ == "__main__":
    main()
