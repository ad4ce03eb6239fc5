"""Micro-benchmark of hcir_attn_fwd at the ViT-B/16 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib
L = _lib.lib()
b, t, h = (int(sys.argv[1]) if len(sys.argv) > 1 else 220), 197, 12
qkv = (torch.randn(b, t, 3, h, 64, device="cuda")).half()
out = torch.empty(b, t, h * 64, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
f = lambda: L.hcir_attn_fwd(qkv.data_ptr(), b, t, h, 64, 0.125, t, out.data_ptr(), st)
for _ in range(3): assert f() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"attn b={b} t={t} h={h}: {ms*1e3:.1f} us  {4*b*h*t*t*64/ms/1e9:.1f} TFLOP/s (unpadded flops)")
