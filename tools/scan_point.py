"""The scan at its 64-query streaming point, alone (for `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` passes):
1 M x 768 fp16 gallery, top-16 of 64 queries, 6 calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import ops
g = torch.nn.functional.normalize(torch.randn(1_000_000, 768, device="cuda"), dim=1).half()
q = torch.nn.functional.normalize(torch.randn(64, 768, device="cuda"), dim=1).half()
for _ in range(6):
    ops.sim_topk(q, g, 16)
torch.cuda.synchronize()
