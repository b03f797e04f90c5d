#!/usr/bin/env python3
"""Aggregate the rocprofv3 PMC passes of tools/pmc_step.sh into profiles/<tag>_pmc_summary.json:
fabric-side bytes per launch of every kernel (FETCH_SIZE doubled per the gfx950 correction of
MI355X_MICROARCH.md, + WRITE_SIZE; both counters are in KiB and are L2->fabric requests, so
Infinity-Cache hits are included).  python tools/pmc_summary.py <gpurun_out dir> <tag> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


_EPI_TAIL = {"1": ",dgrad_bn", "2": ",dgrad_bn", "3": ",affine_elu", "4": ",affine_elu", "5": ",affine_elu",
             "6": ",affine_elu"}


_V2_EPI = {"0": "plain", "1": "dgrad_bn", "3": "affine_elu", "4": "affine_elu", "5": "affine_elu", "6": "affine_elu"}


def key_of(name):
    if "gemm_bf16_v2rc_kernel" in name:     # the weight gradients on the 4-wave loop (SPLIT = false in the bf16 step)
        return "gemm_bf16_v2rc_kernel<f32>" if ("Lb0E" in name or "<false>" in name) else "gemm_bf16_v2rc_kernel<f32,split3>"
    # round 4: the 4-wave tile loop, (anonymous namespace)::v2::gemm_bf16_v2_kernel<TC, EPI, SPLIT> -> ops.py's keys
    m = re.match(r"_ZN12_GLOBAL__N_12v219gemm_bf16_v2_kernelI(DF16b|f)Li(\d)ELb(\d)E", name)
    if m is None and "gemm_bf16_v2_kernel<bool _Accum, int, E, false, false>" in name:
        # rocprofv3's demangler garbles this one instantiation's argument list; in the train step it is the fused dgrad
        # (<__bf16, EPI_DGRAD_BN, false, false>: the only v2 instantiation besides the forward's, which stays mangled)
        return "gemm_bf16_v2_kernel<bf16,dgrad_bn>"
    if m is None:
        n2 = name.replace("(anonymous namespace)::", "").replace("void ", "")
        m2 = re.match(r"v2::gemm_bf16_v2_kernel<(__bf16|float), (\d), (false|true)>", n2)
        if m2:
            m = (m2.group(1) == "__bf16" and "DF16b" or "f", m2.group(2), "1" if m2.group(3) == "true" else "0")
    else:
        m = m.groups()
    if m:
        epi = _V2_EPI.get(m[1], "plain")
        if m[2] == "1":
            return "gemm_bf16_v2_kernel<f32,split3>" if epi == "plain" else "gemm_bf16_v2_kernel<f32,split3," + epi + ">"
        dt = "bf16" if (m[0] == "DF16b" or epi == "affine_elu") else "f32"
        return f"gemm_bf16_v2_kernel<{dt},{epi}>"
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:                                   # rocprofv3 leaves some template instantiations mangled
        ln = int(m.group(1))
        base, rest = name[m.end():m.end() + ln], name[m.end() + ln:]
        if base == "gemm_bf16_dma_kernel":
            # template <TC, ALAY, BLAY, EPI, MF>; EPI 1/2 = dgrad fused with the BatchNorm backward, 3..6 = eval epilogues
            t = re.match(r"I(DF16b|f)Li(\d)ELi(\d)ELi(\d+)E", rest)
            lay = "KC" if t.group(2) == "0" else "RC"
            tail = _EPI_TAIL.get(t.group(4), "")
            return f"gemm_bf16_dma_kernel<{'bf16' if (t.group(1) == 'DF16b' or tail == ',affine_elu') else 'f32'},{lay},{lay}{tail}>"
        return base + ("<bf16>" if rest.startswith("IDF16b") else "")
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"gemm_bf16_dma_kernel<(__bf16|float), (\d), (\d), (\d+)", n)
    if m:
        lay = "KC" if m.group(2) == "0" else "RC"
        tail = _EPI_TAIL.get(m.group(4), "")
        return f"gemm_bf16_dma_kernel<{'bf16' if (m.group(1) == '__bf16' or tail == ',affine_elu') else 'f32'},{lay},{lay}{tail}>"
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"<.*", "", n)
    return n


def collect(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = agg[key_of(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg


def main():
    root, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch = collect(f"{root}/pmc_{tag}_fetch", "FETCH_SIZE")
    write = collect(f"{root}/pmc_{tag}_write", "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, (0, 0.0))
        nw, vw = write.get(k, (0, 0.0))
        fb = 2.0 * 1024.0 * vf / max(nf, 1)
        wb = 1024.0 * vw / max(nw, 1)
        if fb + wb < 8e6:
            continue                       # keep the file small: only kernels that move >= 8 MB per launch
        kernels[k] = {"launches_profiled": max(nf, nw), "fetch_bytes_per_launch_corrected": fb,
                      "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    doc = {
        "source": "tools/pmc_step.sh: rocprofv3 --kernel-trace --pmc <C> -- python bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-kernel-timing (separate passes for FETCH_SIZE and WRITE_SIZE)",
        "workload": "B=64 T=30 N=128 C=4 K=8 bf16",
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled; "
                      "WRITE_SIZE exact; both in KiB; both are L2->fabric requests, Infinity-Cache hits included "
                      "(MI355X_MICROARCH.md HBM section)",
        "kernels": kernels,
    }
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k:48s} n={v['launches_profiled']:3d} fetch {v['fetch_bytes_per_launch_corrected'] / 1e6:9.1f} MB "
              f"write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
