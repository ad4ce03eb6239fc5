"""A/B of libhcir builds at the ViT-B/16 GEMM shapes, interleaved rounds in ONE process
(cdna_hip_programming.md §5.4 rule 24): every variant is a separately built .so of the same sources.

usage: python3 tools/ab_gemm.py <batch> tag=path [tag=path ...] [zeros]
       (tag 'base' = the in-tree library when no path is given: base=)
Each shape: `rounds` rounds of {variant A x iters, variant B x iters, ...}; prints median and min us per variant.
"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib


def load(path):
    l = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
    return l


def main():
    b = int(sys.argv[1])
    zeros = "zeros" in sys.argv[2:]      # zero-filled operands: the clock the chip holds without data toggling
    libs = []
    for spec in [a for a in sys.argv[2:] if a != "zeros"]:
        tag, _, path = spec.partition("=")
        libs.append((tag, load(path or _lib.LIB_PATH)))
    m = b * 197
    st = torch.cuda.current_stream().cuda_stream
    shapes = ((2304, 768, 0, "qkv"), (768, 768, 6, "proj"), (3072, 768, 1, "fc1"), (768, 3072, 6, "fc2"))
    rounds, iters = 7, 10
    tot = {t: 0.0 for t, _ in libs}
    for (n, k, epi, name) in shapes:
        a = (torch.randn(m, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
        bias = torch.randn(n, device="cuda")
        if zeros:
            a.zero_(), w.zero_(), bias.zero_()
        out = torch.zeros(m, n, device="cuda", dtype=torch.float16)
        call = lambda L: L.hcir_gemm_f16(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, epi,
                                         out.data_ptr(), n, st)
        for _, L in libs:                       # warm-up (clock ramp: the first ~50 launches run slower)
            for _ in range(30):
                assert call(L) == 0
        times = {t: [] for t, _ in libs}
        for r in range(rounds):
            for t, L in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    call(L)
                e1.record()
                torch.cuda.synchronize()
                times[t].append(e0.elapsed_time(e1) / iters * 1e3)
        for t, _ in libs:
            med, mn = statistics.median(times[t]), min(times[t])
            tot[t] += med
            print(f"{name:5s} M={m} N={n} K={k} {t:10s} median {med:8.1f} us  min {mn:8.1f} us  "
                  f"{2*m*n*k/med/1e6:7.1f} TF", flush=True)
    fl = 2 * m * 768 * (2304 + 768 + 3072 + 3072)
    for t, _ in libs:
        print(f"layer {t:10s} {tot[t]:8.1f} us  {fl/tot[t]/1e6:7.1f} TF")


if __name__ == "__main__":
    main()
