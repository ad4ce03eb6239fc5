"""Micro-benchmark of hcir_sim_topk alone (GB/s of gallery streamed, TFLOP/s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import ops

def run(nq, ng, d, k, dtype, iters=10):
    g = torch.randn(ng, d, device="cuda")
    g = g / g.norm(dim=1, keepdim=True)
    q = torch.randn(nq, d, device="cuda")
    q = q / q.norm(dim=1, keepdim=True)
    g, q = g.to(dtype), q.to(dtype)
    for _ in range(2):
        ops.sim_topk(q, g, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.sim_topk(q, g, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gb = ng * d * g.element_size() / 1e9
    print(f"{str(dtype):16s} nq={nq:4d} ng={ng:8d} d={d} k={k}: {ms:8.3f} ms  "
          f"{gb / ms * 1e3:8.1f} GB/s  {2 * nq * ng * d / ms / 1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    ng = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    for dtype in (torch.float32, torch.float16):
        for nq in (1, 32, 64, 128, 220, 256, 512, 1760):
            run(nq, ng, 768, 10 if nq != 220 else 16, dtype)
    run(64, 10_000, 768, 10, torch.float32)
    run(64, 10_000, 768, 10, torch.float16)
