"""Where an attention-backward workgroup spends its time: run against a library built with -DHCIR_ATTN_BWD_STAMPS
(tools/build_variant.sh abstamps "-DHCIR_ATTN_BWD_STAMPS" attn_bwd.hip).  Wave 0 of workgroup 0 stamps the phase
boundaries of its fourth item with (s_memtime shader cycles, s_memrealtime 100 MHz).
usage: python3 tools/diag_attn_bwd.py [batch=3072] [lib=tools/_libhcir_abstamps.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib

b = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "_libhcir_abstamps.so")
L = ctypes.CDLL(path)
for name in ("hcir_attn_bwd", "hcir_attn_fwd_lse"):
    fn = getattr(L, name)
    fn.restype, fn.argtypes = _lib.SIGNATURES[name]
t, h = 197, 12
g = torch.Generator(device="cuda").manual_seed(3)
qkv = (torch.randn(b, t, 3, h, 64, device="cuda", generator=g) * 0.8).half()
dout = torch.randn(b, t, h * 64, device="cuda", generator=g).half()
out = torch.empty(b, t, h * 64, device="cuda", dtype=torch.float16)
lse = torch.empty(b, h, t, device="cuda", dtype=torch.float32)
dqkv = torch.empty_like(qkv)
st = torch.cuda.current_stream().cuda_stream
assert L.hcir_attn_fwd_lse(qkv.data_ptr(), b, t, h, 64, 0.125, out.data_ptr(), lse.data_ptr(), st) == 0
names = ["loop top -> R0 landed + barrier", "dQ(prev) stores, D, barrier", "pass 1 (7 query tiles, K/V transfers inside)",
         "pass-2 operands, dK/dV packing, R1 landed + barrier", "pass 2 (7 key tiles; stores, next R0 inside)"]
idx = [0, 1, 2, 3, 4, 6]
for rep in range(4):
    assert L.hcir_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), b, t, h, 64, 0.125,
                           dqkv.data_ptr(), st) == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert L.hcir_diag_attn_bwd_stamps(buf) == 0
    cyc = [buf[2 * i] for i in idx]
    rt = [buf[2 * i + 1] for i in idx]
    if rep < 2:
        continue
    tot_c, tot_us = cyc[-1] - cyc[0], (rt[-1] - rt[0]) / 100.0
    print(f"run {rep}: item {tot_c} cycles = {tot_us:.2f} us  ({tot_c / max(tot_us, 1e-9) / 1e3:.2f} GHz)")
    for i, n in enumerate(names):
        print(f"   {n:58s} {cyc[i + 1] - cyc[i]:7d} cycles")
