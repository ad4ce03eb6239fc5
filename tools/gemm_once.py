"""One launch set of the four ViT-B/16 GEMM shapes (for rocprofv3 --kernel-trace --stats): python3 tools/gemm_once.py <batch> [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib
L = _lib.lib()
b = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = b * 197
st = torch.cuda.current_stream().cuda_stream
for (n, k, epi) in ((2304, 768, 0), (768, 768, 6), (3072, 768, 1), (768, 3072, 6)):
    a = (torch.randn(m, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.randn(n, device="cuda")
    out = torch.zeros(m, n, device="cuda", dtype=torch.float16)
    for _ in range(iters):
        assert L.hcir_gemm_f16(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, epi, out.data_ptr(), n, st) == 0
torch.cuda.synchronize()
