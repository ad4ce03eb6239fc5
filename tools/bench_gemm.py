"""Micro-benchmark of hcir_gemm_f16 at the ViT-B/16 shapes (M = batch*197)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib

def run(L, m, n, k, epi, iters=20):
    a = (torch.randn(m, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.randn(n, device="cuda")
    out = torch.zeros(m, n, device="cuda", dtype=torch.float32 if epi in (2, 3) else torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: L.hcir_gemm_f16(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, epi, out.data_ptr(), n, st)
    for _ in range(3):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={m} N={n} K={k} epi={epi}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.1f} TFLOP/s", flush=True)
    return ms

if __name__ == "__main__":
    # HCIR_LIBVAR=<tag> loads tools/_libhcir_<tag>.so instead (ablation builds of the same sources)
    if os.environ.get("HCIR_LIBVAR"):
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "_libhcir_%s.so" % os.environ["HCIR_LIBVAR"])
    L = _lib.lib()
    # the first ~50 launches of a process run 10-17 % slower (clock ramp): warm up before timing
    run(L, 220 * 197, 2304, 768, 0, iters=100)
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    m = b * 197
    tot = 0
    for (n, k, epi, name) in ((2304, 768, 0, "qkv"), (768, 768, 2, "proj"), (3072, 768, 1, "fc1"), (768, 3072, 2, "fc2")):
        tot += run(L, m, n, k, epi)
    fl = 2 * m * 768 * (2304 + 768 + 3072 + 3072)
    print(f"layer total {tot*1e3:.1f} us  {fl/tot/1e9:.1f} TFLOP/s")
    run(L, 8192, 8192, 8192, 0, iters=5)
