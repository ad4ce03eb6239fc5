import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import ops
def run(nq, ng, d, k):
    g = torch.nn.functional.normalize(torch.randn(ng, d, device="cuda"), dim=1).half()
    q = torch.nn.functional.normalize(torch.randn(nq, d, device="cuda"), dim=1).half()
    for _ in range(3): ops.sim_topk(q, g, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.sim_topk(q, g, k)
    e1.record(); torch.cuda.synchronize()
    print(f"nq={nq} ng={ng} d={d} k={k}: {e0.elapsed_time(e1)/10*1e3:.1f} us", flush=True)
for k in (10, 20, 50):
    for ng in (256, 4096, 30000):
        run(32, ng, 1024, k)
