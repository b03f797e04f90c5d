import torch, sys, os
sys.path.insert(0, os.getcwd())
from opensetgaitrecognition_pcaa_amd import ops
from opensetgaitrecognition_pcaa_amd._lib import KC, PCAA_BF16
P = 245760
for cin, cout in ((512, 512), (512, 1024), (1024, 1024)):
    x = (torch.randn(P, cin, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(cout, cin, device="cuda") * 0.05).bfloat16()
    y = torch.empty(P, cout, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * P * cin * cout
    def vend(): torch.matmul(x, w.t(), out=y)
    def ours(): ops.gemm(x, KC, w, KC, P, cout, cin, out=y, out_dtype=torch.bfloat16, math=PCAA_BF16)
    for name, fn in (("hipBLASLt", vend), ("ours(plain)", ours)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        single = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            single.append(e0.elapsed_time(e1))
        single.sort()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        batch = e0.elapsed_time(e1) / 20
        print(f"[{cin}->{cout}] {name:12s} single median {single[10]:.3f} ms ({fl/single[10]/1e9:.0f} TF)  min {single[0]:.3f}   batch-of-20 {batch:.3f} ms ({fl/batch/1e9:.0f} TF)")
