"""The scan at a streaming point, alone (for `rocprofv3 --kernel-trace` / `--pmc FETCH_SIZE / WRITE_SIZE` passes).
usage: scan_point.py [c4|c5] [nq]
  c4: 1 M x 768 fp16 gallery, top-16 (the filter scan of the timed path), default 64 queries
  c5: 1.25 M x 1024 fp16 shard, top-50 (BASELINE.json configs[4]), default 32 queries
6 calls each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import ops
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
ng, d, k, nq = (1_000_000, 768, 16, 64) if cfg == "c4" else (1_250_000, 1024, 50, 32)
if len(sys.argv) > 2:
    nq = int(sys.argv[2])
g = torch.empty(ng, d, device="cuda", dtype=torch.float16)
for s in range(0, ng, 250_000):
    g[s:s + 250_000] = torch.nn.functional.normalize(torch.randn(min(250_000, ng - s), d, device="cuda"), dim=1).half()
q = torch.nn.functional.normalize(torch.randn(nq, d, device="cuda"), dim=1).half()
for _ in range(6):
    ops.sim_topk(q, g, k)
torch.cuda.synchronize()
