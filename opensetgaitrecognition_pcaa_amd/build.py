"""Build recipe for the HIP extension: ``hipcc --offload-arch=gfx950`` on every
``csrc/*.hip`` -> one in-tree ``libpcaa_hip.so`` (C ABI, include/pcaa_hip.h).
Cross-compiles without a GPU.  ``python -m opensetgaitrecognition_pcaa_amd.build``.
"""
import glob
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libpcaa_hip.so")
STAMP = os.path.join(PKG, "csrc", ".build_stamp")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
# lab builds only (same-box A/B of a compile-time variant, e.g. PCAA_HIPCC_EXTRA=-DPCAA_V2_YRING=2); part of the digest
FLAGS += os.environ.get("PCAA_HIPCC_EXTRA", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the PCAA HIP extension cannot be built")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    files = _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(os.path.dirname(PKG), "include", "pcaa_hip.h")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in _sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"hipcc failed on {src}\n{out.decode(errors='replace')}\n")
    if failed:
        raise RuntimeError("building libpcaa_hip.so failed")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
