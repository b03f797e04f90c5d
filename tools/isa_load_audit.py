"""Scan the gfx950 ISA of the HIP sources (hipcc -save-temps) for global loads that are waited for one at a time.

A load that sits under a divergent branch together with its first use is followed by ``s_waitcnt vmcnt(0)`` inside that
branch: N such loads in a loop run as N serialised round trips to HBM instead of N requests in flight.  Round 3 found
the first PointNet layer's backward passes doing exactly that (0.113 -> 0.078 ms once the loads were made unconditional
on clamped rows).  This lists, per kernel, the loop bodies that hold two or more ``s_waitcnt vmcnt(0)`` -- candidates,
not verdicts: a loop that loads, waits and stores once per trip with 32 waves per CU behind it is fine (the
column-invariant elementwise kernels: unrolling them 2/4/8x changed nothing, profiles/r03_elementwise_unroll_lab.txt).

    cd opensetgaitrecognition_pcaa_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -c X.hip -o /tmp/X.o -save-temps=obj
    python tools/isa_load_audit.py /tmp/X-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()[:110]
    except Exception:
        return n


def audit(path):
    s = open(path).read()
    for m in re.finditer(r"^(\S+):\s*; @\S+\n", s, re.M):
        name = m.group(1)
        i = m.end()
        j = s.find(".Lfunc_end", i)
        body = s[i:j].split("\n")
        # loops: a block labelled "Loop Header" ... up to the backward branch to it; approximate by scanning label ranges
        labels = {}
        for k, l in enumerate(body):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = k
        loops = []
        for k, l in enumerate(body):
            mm = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                loops.append((labels[mm.group(1)], k))
        rows = []
        for a, b in loops:
            ins = [x.strip() for x in body[a:b + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            loads = [x for x in ins if x.startswith(("global_load", "buffer_load", "flat_load"))]
            waits = [x for x in ins if "vmcnt" in x]
            if len(loads) >= 2:
                z = sum(1 for x in waits if "vmcnt(0)" in x)
                rows.append((len(ins), len(loads), len(waits), z))
        bad = [r for r in rows if r[3] >= 2]
        if bad:
            print(demangle(name))
            for r in bad:
                print("    loop of %4d instr: %2d loads, %2d vmcnt waits, %d of them vmcnt(0)" % r)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print("==", p)
        audit(p)
