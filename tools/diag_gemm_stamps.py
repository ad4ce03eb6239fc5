"""Diagnostic: wall-clock stamps around the second tile of every workgroup of the 256 x 256 GEMM kernel, from a library
built with -DHCIR_DIAG_GSTAMPS (tools/build_variant.sh gstamps "-DHCIR_DIAG_GSTAMPS" gemm.hip).
usage: diag_gemm_stamps.py tools/_libhcir_gstamps.so [zeros]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import numpy as np
import torch
from hcir import _lib

L = ctypes.CDLL(sys.argv[1])
fn = L.hcir_gemm_f16
fn.restype, fn.argtypes = _lib.SIGNATURES["hcir_gemm_f16"]
M = 880 * 197
ZEROS = "zeros" in sys.argv[2:]
st = torch.cuda.current_stream().cuda_stream
buf = np.zeros(256 * 8, dtype=np.uint64)
names = ["main loop of the tile (nkc k-steps)", "barrier before the epilogue", "epilogue", "epilogue end -> next tile's stage 0 ready",
         "first k-step of the next tile"]
OV = "ov" in sys.argv[2:]     # stamp layout of gemm_f16_ov_kernel (qkv / fc1): the boundary step
if OV:
    names = ["k-steps 1 .. nkc-1 of the tile", "boundary: wait for stage 0 + vectors to LDS + barrier", "boundary: issue of the next stage (8 transfers)",
             "boundary: 8 sub-passes (finish, MFMA, store)", "next step's wait (vmcnt 16) + barrier"]
for (tag, n, k, epi) in (("qkv  N=2304 K=768  BIAS_F16", 2304, 768, 0), ("fc1  N=3072 K=768  BIAS_GELU_F16", 3072, 768, 1),
                         ("proj N=768  K=768  BIAS_RESID_F16", 768, 768, 6), ("fc2  N=768  K=3072 BIAS_RESID_F16", 768, 3072, 6)):
    a = (torch.randn(M, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * 0.03).half()
    bias = torch.randn(n, device="cuda")
    if ZEROS:
        a.zero_(), w.zero_(), bias.zero_()
    out = torch.zeros(M, n, device="cuda", dtype=torch.float16)
    for it in range(40):   # clock settles over the first tens of launches
        assert fn(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, M, n, k, epi, out.data_ptr(), n, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(10):
        fn(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, M, n, k, epi, out.data_ptr(), n, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    assert L.hcir_debug_gemm_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
    s = buf.reshape(256, 8).astype(np.int64)
    print(f"{tag}: {ms * 1e3:.0f} us per launch, {2.0 * M * n * k / ms / 1e9:.0f} TFLOP/s; tiles per workgroup {M / 256 * n / 256 / 256:.1f}")
    for i, nm in enumerate(names if epi in (0, 1) or not OV else []):
        dlt = (s[:, i + 1] - s[:, i]) / 100.0
        print(f"   {nm:42s} median {np.median(dlt):6.2f} us   p10 {np.percentile(dlt, 10):6.2f}   p90 {np.percentile(dlt, 90):6.2f}")
    if OV and epi in (0, 1) and hasattr(L, "hcir_debug_gemm_wstamps"):
        wb = np.zeros(256 * 24, dtype=np.uint64)
        assert L.hcir_debug_gemm_wstamps(wb.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
        w = wb.reshape(256, 3, 8).astype(np.int64)
        t0 = w[:, 0, :].min(axis=1, keepdims=True)
        print("   per wavefront, us after the first wavefront entered the boundary's sub-passes (median over workgroups):")
        print("     wavefront        " + " ".join(f"{i:6d}" for i in range(8)))
        for nm, k in (("enters sub-passes", 0), ("leaves sub-passes", 1), ("next step starts ", 2)):
            print(f"     {nm}" + " ".join(f"{np.median((w[:, k, i:i+1] - t0) / 100.0):6.2f}" for i in range(8)))
    last = 5 if OV and epi in (0, 1) else 4
    print(f"   {'tile period':42s} median {np.median((s[:, last] - s[:, 0]) / 100.0):6.2f} us")
    clk = (s[:, 7] - s[:, 6]) / np.maximum(s[:, last] - s[:, 0], 1) * 100.0
    print(f"   {'in-kernel shader clock over that tile':42s} median {np.median(clk):6.0f} MHz  p10 {np.percentile(clk, 10):6.0f}  p90 {np.percentile(clk, 90):6.0f}")
