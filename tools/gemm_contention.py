#!/usr/bin/env python3
"""The PointNet forward GEMM beside a kernel that holds H CUs for its whole duration (tools/microbench/hog.hip: stands
in for a collective's persistent workgroups): how the tile loop copes when it does not get every CU.
    python tools/gemm_contention.py            (needs tools/microbench/libhog.so, see hog.hip)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opensetgaitrecognition_pcaa_amd import ops, _lib
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16
hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "microbench", "libhog.so"))
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
P, cin, cout = 245760, 1024, 1024
x = (torch.randn(P, cin, device="cuda") * 0.5).bfloat16()
w = (torch.randn(cout, cin, device="cuda") * 0.05).bfloat16()
y = torch.empty(P, cout, device="cuda", dtype=torch.bfloat16)
out = torch.zeros(4, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
fl = 2.0 * P * cin * cout
for variant in ((0, "tile tickets"),):
    for H in (0, 8, 16, 32, 64):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            if H:
                with torch.cuda.stream(side):
                    hog.hog_launch(H, 600, out.data_ptr(), side.cuda_stream)     # ~2-3 ms
                torch.cuda._sleep(200000)                                        # let the hog get its CUs first
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(x, KC, w, KC, P, cout, cin, out=y, out_dtype=torch.bfloat16, math=PCAA_BF16)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"{variant[1]:14s} {H:3d} CUs held: GEMM {ts[2]:.3f} ms ({fl / ts[2] / 1e9:.0f} TF; ideal with {256 - H} CUs: x{256 / (256 - H):.2f})")
