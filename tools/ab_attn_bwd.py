"""A/B of libhcir builds on hcir_attn_bwd at the ViT-B/16 shape (batch argv[1], the training step runs 3 x 1024 rows),
interleaved rounds in one process; results compared with the first library's (max abs difference, relative to max |d|).
usage: python3 tools/ab_attn_bwd.py <batch> tag=path [tag=path ...]   ('base=' = the in-tree library)"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib


def load(path):
    l = ctypes.CDLL(path)
    for name in ("hcir_attn_bwd", "hcir_attn_fwd_lse"):
        fn = getattr(l, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    return l


b, t, h = int(sys.argv[1]), 197, 12
libs = []
for spec in sys.argv[2:]:
    tag, _, path = spec.partition("=")
    libs.append((tag, load(path or _lib.LIB_PATH)))
g = torch.Generator(device="cuda").manual_seed(3)
qkv = (torch.randn(b, t, 3, h, 64, device="cuda", generator=g) * 0.8).half()
dout = torch.randn(b, t, h * 64, device="cuda", generator=g).half()
out = torch.empty(b, t, h * 64, device="cuda", dtype=torch.float16)
lse = torch.empty(b, h, t, device="cuda", dtype=torch.float32)
dqkv = torch.empty_like(qkv)
st = torch.cuda.current_stream().cuda_stream
assert libs[0][1].hcir_attn_fwd_lse(qkv.data_ptr(), b, t, h, 64, 0.125, out.data_ptr(), lse.data_ptr(), st) == 0
call = lambda L: L.hcir_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), b, t, h, 64, 0.125,
                                 dqkv.data_ptr(), st)
ref = None
for tag, L in libs:
    dqkv.fill_(float("nan"))
    for _ in range(3):
        assert call(L) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv).all(), tag
    if ref is None:
        ref = dqkv.float().clone()
    else:
        d = (dqkv.float() - ref).abs().max().item()
        print(f"{tag}: max |diff| vs {libs[0][0]} = {d:.3e} (max |d| {ref.abs().max().item():.3e})")
    again = dqkv.clone()
    call(L)
    torch.cuda.synchronize()
    assert torch.equal(again, dqkv), f"{tag}: not deterministic"
times = {tag: [] for tag, _ in libs}
for r in range(7):
    for tag, L in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call(L)
        e1.record()
        torch.cuda.synchronize()
        times[tag].append(e0.elapsed_time(e1) / 10 * 1e3)
gb = (2 * qkv.numel() + 2 * out.numel()) * 2 / 1e3
for tag, _ in libs:
    med = statistics.median(times[tag])
    print(f"attn_bwd b={b}: {tag:10s} median {med:8.1f} us  min {min(times[tag]):8.1f} us  {gb / med:6.0f} GB/s  "
          f"{10 * b * h * t * t * 64 / med / 1e6:6.1f} TFLOP/s (10 T^2 64 per head)")
