"""A/B of libhcir builds on hcir_attn_fwd at the ViT-B/16 shape (batch argv[1]), interleaved rounds in one process.
usage: python3 tools/ab_attn.py <batch> tag=path [tag=path ...]   ('base=' = the in-tree library)"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib


def load(path):
    l = ctypes.CDLL(path)
    fn = l.hcir_attn_fwd
    fn.restype, fn.argtypes = _lib.SIGNATURES["hcir_attn_fwd"]
    return l


b, t, h = int(sys.argv[1]), 197, 12
libs = []
for spec in sys.argv[2:]:
    tag, _, path = spec.partition("=")
    libs.append((tag, load(path or _lib.LIB_PATH)))
qkv = torch.randn(b, t, 3, h, 64, device="cuda").half()
out = torch.empty(b, t, h * 64, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
call = lambda L: L.hcir_attn_fwd(qkv.data_ptr(), b, t, h, 64, 0.125, t, out.data_ptr(), st)
ref = None
for tag, L in libs:
    for _ in range(5):
        assert call(L) == 0
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    else:
        assert torch.equal(ref, out), f"{tag}: results differ"
times = {tag: [] for tag, _ in libs}
for r in range(7):
    for tag, L in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call(L)
        e1.record()
        torch.cuda.synchronize()
        times[tag].append(e0.elapsed_time(e1) / 20 * 1e3)
gb = (qkv.numel() + out.numel()) * 2 / 1e3
for tag, _ in libs:
    med = statistics.median(times[tag])
    print(f"attn b={b}: {tag:10s} median {med:7.1f} us  min {min(times[tag]):7.1f} us  {gb / med:6.0f} GB/s  "
          f"{4 * b * h * t * t * 64 / med / 1e6:6.1f} TFLOP/s")
