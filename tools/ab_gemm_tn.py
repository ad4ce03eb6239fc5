"""A/B of libhcir builds on hcir_gemm_f16_tn at the four weight-gradient shapes of a ViT-B/16 block (M = batch x 197
rows padded to 64; the training step runs 3 x 1024 images), interleaved rounds in one process, results compared bit for
bit with the first library's.   usage: python3 tools/ab_gemm_tn.py <batch> tag=path [tag=path ...]  ('base=' in-tree)"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib


def load(path):
    l = ctypes.CDLL(path)
    for name in ("hcir_gemm_f16_tn", "hcir_gemm_f16_tn_workspace_bytes"):
        fn = getattr(l, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    return l


batch = int(sys.argv[1])
m = (batch * 197 + 63) // 64 * 64
libs = []
for spec in sys.argv[2:]:
    tag, _, path = spec.partition("=")
    libs.append((tag, load(path or _lib.LIB_PATH)))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(1)
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
total = {tag: 0.0 for tag, _ in libs}
for name, n, k in shapes:
    a = torch.randn(m, n, device="cuda", generator=g).half()
    b = torch.randn(m, k, device="cuda", generator=g).half()
    dw = torch.empty(n, k, device="cuda")
    wsb = libs[0][1].hcir_gemm_f16_tn_workspace_bytes(m, n, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    call = lambda L: L.hcir_gemm_f16_tn(a.data_ptr(), n, b.data_ptr(), k, m, n, k, dw.data_ptr(), k, 0, ws.data_ptr(), wsb, st)
    ref = None
    for tag, L in libs:
        for _ in range(3):
            assert call(L) == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = dw.clone()
        else:
            assert torch.equal(ref, dw), f"{name} {tag}: results differ"
    times = {tag: [] for tag, _ in libs}
    for r in range(7):
        for tag, L in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call(L)
            e1.record()
            torch.cuda.synchronize()
            times[tag].append(e0.elapsed_time(e1) / 5 * 1e3)
    for tag, _ in libs:
        med = statistics.median(times[tag])
        total[tag] += med
        print(f"gemm_tn {name:4s} M={m} N={n} K={k}: {tag:8s} median {med:8.1f} us  {2.0 * m * n * k / med / 1e6:7.1f} TFLOP/s", flush=True)
    del a, b, dw, ws
flops = sum(2.0 * m * n * k for _, n, k in shapes)
for tag, _ in libs:
    print(f"block total: {tag:8s} {total[tag]:9.1f} us  {flops / total[tag] / 1e6:7.1f} TFLOP/s")
