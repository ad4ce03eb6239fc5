"""C5's similarity scan alone (top-50 over 1.25 M x 1024 fp16, 32 and 64 queries), for rocprofv3 --kernel-trace --stats
and for HIP-event timing: python3 tools/prof_c5.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
import torch.nn.functional as F
from hcir import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
ng, d, k = 1_250_000, 1024, 50
g = torch.empty(ng, d, device=dev, dtype=torch.float16)
gen = torch.Generator(device=dev).manual_seed(2000)
for s in range(0, ng, 250_000):
    g[s:s + 250_000] = F.normalize(torch.randn(250_000, d, device=dev, generator=gen), dim=1).half()
q = F.normalize(torch.randn(64, d, device=dev, generator=gen), dim=1).half()
for nq in (32, 64):
    qq = q[:nq].contiguous()
    for _ in range(5):
        ops.sim_topk(qq, g, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.sim_topk(qq, g, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gbs = (ng * d * 2 + nq * d * 2 + nq * k * 12) / (ms * 1e-3) / 1e9
    print(f"top-{k}, {nq} queries: {ms * 1e3:.1f} us  {gbs:.0f} GB/s  {gbs / 8000:.3f} of 8 TB/s", flush=True)
