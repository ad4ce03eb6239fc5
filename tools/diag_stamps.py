"""Diagnostic: wall-clock stamps inside the floorless (prefix) launch of hcir_sim_topk, from a library built with
-DHCIR_DIAG_STAMPS (tools/build_variant.sh stamps "-DHCIR_DIAG_STAMPS" sim_topk.hip).
usage: diag_stamps.py tools/_libhcir_stamps.so [c4|c5] [nq]"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import numpy as np
import torch
from hcir import _lib

L = ctypes.CDLL(sys.argv[1])
for name in ("hcir_sim_topk_workspace_bytes", "hcir_sim_topk"):
    fn = getattr(L, name)
    fn.restype, fn.argtypes = _lib.SIGNATURES[name]
cfg = sys.argv[2] if len(sys.argv) > 2 else "c4"
ng, d, k, nq = (1_000_000, 768, 16, 64) if cfg == "c4" else (1_250_000, 1024, 50, 32)
if len(sys.argv) > 3:
    nq = int(sys.argv[3])
g = torch.empty(ng, d, device="cuda", dtype=torch.float16)
for s in range(0, ng, 250_000):
    g[s:s + 250_000] = torch.nn.functional.normalize(torch.randn(min(250_000, ng - s), d, device="cuda"), dim=1).half()
q = torch.nn.functional.normalize(torch.randn(nq, d, device="cuda"), dim=1).half()
val = torch.empty(nq, k, device="cuda")
idx = torch.empty(nq, k, dtype=torch.int64, device="cuda")
ws = torch.empty(L.hcir_sim_topk_workspace_bytes(nq, ng, d, k, 1), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
buf = np.zeros(1024 * 8, dtype=np.uint64)
names = ["entry->prologue", "prologue->stage0 landed", "stage0->last mfma", "last mfma->epilogue end", "in-workgroup merge"]
for it in range(8):
    L.hcir_sim_topk(q.data_ptr(), nq, g.data_ptr(), ng, d, k, 1, None, None, 0, val.data_ptr(), idx.data_ptr(),
                    ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    if it < 5:
        continue
    assert L.hcir_debug_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
    s = buf.reshape(1024, 8).astype(np.int64)
    live = s[:, 0] > 0
    s = s[live]
    print(f"{cfg} nq={nq}: {len(s)} workgroups stamped; launch span (first entry -> last exit) {(s[:, 5].max() - s[:, 0].min()) / 100:.1f} us; "
          f"entry skew {(s[:, 0].max() - s[:, 0].min()) / 100:.1f} us")
    for i, n in enumerate(names):
        dlt = (s[:, i + 1] - s[:, i]) / 100.0
        print(f"   {n:28s} median {np.median(dlt):6.2f} us   max {dlt.max():6.2f} us")
    tot = (s[:, 5] - s[:, 0]) / 100.0
    print(f"   {'workgroup total':28s} median {np.median(tot):6.2f} us   max {tot.max():6.2f} us")
    buf[:] = 0
