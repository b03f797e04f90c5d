#!/usr/bin/env python3
"""Which stream operations of the emulated / real data-parallel step survive a hipGraph capture on this image
(round 6: the first capture of the emulated step died inside hipStreamEndCapture).  Each case runs in a child process."""
import subprocess, sys

CASES = {
    "side_stream_kernel": """
s = torch.cuda.Stream(); x = torch.ones(1 << 20, device='cuda')
g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s):
        s.wait_event(e); x.mul_(2.0); d = torch.cuda.Event(); d.record(s)
    torch.cuda.current_stream().wait_event(d); y = x + 1
g.replay(); torch.cuda.synchronize(); print(float(y[0]))
""",
    "side_stream_d2d_copy": """
s = torch.cuda.Stream(); x = torch.ones(1 << 20, device='cuda'); z = torch.zeros(4, 1 << 20, device='cuda')
g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s):
        s.wait_event(e); z[0].copy_(x); d = torch.cuda.Event(); d.record(s)
    torch.cuda.current_stream().wait_event(d); y = z.sum()
g.replay(); torch.cuda.synchronize(); print(float(y))
""",
    "side_stream_expand_copy": """
s = torch.cuda.Stream(); x = torch.ones(1 << 20, device='cuda'); z = torch.zeros(4, 1 << 20, device='cuda')
g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s):
        s.wait_event(e); z[1:].copy_(x.unsqueeze(0).expand(3, -1)); d = torch.cuda.Event(); d.record(s)
    torch.cuda.current_stream().wait_event(d); y = z.sum()
g.replay(); torch.cuda.synchronize(); print(float(y))
""",
    "bf16_record_stream": """
s = torch.cuda.Stream(); x = torch.ones(1 << 20, device='cuda', dtype=torch.bfloat16); z = torch.zeros(4, 1 << 20, device='cuda', dtype=torch.bfloat16)
g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    w = torch.empty(1 << 20, device='cuda', dtype=torch.bfloat16); w.copy_(x)
    e = torch.cuda.Event(); e.record(); w.record_stream(s); z.record_stream(s)
    with torch.cuda.stream(s):
        s.wait_event(e); z.view(-1).view(4, -1)[0].copy_(w); z[1:].copy_(w.unsqueeze(0).expand(3, -1)); d = torch.cuda.Event(); d.record(s)
    torch.cuda.current_stream().wait_event(d); y = z.float().sum()
g.replay(); torch.cuda.synchronize(); print(float(y))
""",
    "rccl_1rank_allreduce_allgather": """
import torch.distributed as dist
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1, device_id=torch.device('cuda', 0))
x = torch.ones(1 << 20, device='cuda'); o = torch.zeros(1 << 20, device='cuda')
dist.all_reduce(x); dist.all_gather_into_tensor(o, x); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    dist.all_reduce(x)
    w = dist.all_gather_into_tensor(o, x, async_op=True); w.wait(); y = o.sum()
g.replay(); torch.cuda.synchronize(); print(float(y)); dist.destroy_process_group()
""",
}
if len(sys.argv) > 1:
    import torch
    exec(CASES[sys.argv[1]])
    sys.exit(0)
for name in CASES:
    r = subprocess.run([sys.executable, __file__, name], capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr).strip().splitlines()[-2:]
    print(f"{name}: rc={r.returncode} {' | '.join(tail)}", flush=True)
