"""Bit-for-bit comparison of two libhcir builds on the persistent GEMM's fp16 epilogues (plain and LayerNorm-folded),
at shapes with several tiles per workgroup and a ragged last row of tiles.

usage: python3 tools/cmp_gemm_variants.py <other.so>      (the in-tree library against <other.so>)
"""
import ctypes, os, sys
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
    A, B = load(_lib.LIB_PATH), load(sys.argv[1])
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for (m, n, k) in ((197 * 100, 2304, 768), (197 * 100, 3072, 768), (197 * 300 + 5, 3072, 768), (197 * 90, 768, 3072),
                      (197 * 880, 3072, 768), (70000, 512, 128), (197 * 150, 1024, 1024)):
        g = torch.Generator(device="cuda").manual_seed(m + n)
        a = (torch.randn(m, k, device="cuda", generator=g) * 0.5 + torch.randn(m, 1, device="cuda", generator=g)).half()
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).half()
        bias = torch.randn(n, device="cuda", generator=g)
        c1 = w.float().sum(1)
        x = a.float()
        stats = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()], 1).contiguous()
        for epi in (0, 1):
            outs = []
            for L in (A, B):
                o = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float16)
                assert L.hcir_gemm_f16(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, epi,
                                       o.data_ptr(), n, st) == 0
                outs.append(o)
            torch.cuda.synchronize()
            eq = torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0]).any().item()
            bad += not eq
            print(f"m={m} n={n} k={k} epi={epi} plain : {'equal' if eq else 'DIFFERENT'}", flush=True)
            if not eq:
                d = (outs[0].float() - outs[1].float()).abs()
                idx = torch.nonzero(d > 0)
                print("   first differing (row, col):", idx[:5].tolist(), "count", idx.shape[0], "max", d.max().item(),
                      "nan", torch.isnan(outs[0]).sum().item())
            outs = []
            for L in (A, B):
                o = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float16)
                assert L.hcir_gemm_f16_fused(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, epi,
                                             o.data_ptr(), n, stats.data_ptr(), c1.data_ptr(), None, st) == 0
                outs.append(o)
            torch.cuda.synchronize()
            eq = torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0]).any().item()
            bad += not eq
            print(f"m={m} n={n} k={k} epi={epi} ln-fold: {'equal' if eq else 'DIFFERENT'}", flush=True)
            if not eq:
                d = (outs[0].float() - outs[1].float()).abs()
                idx = torch.nonzero(~(d == 0))
                print("   first differing (row, col):", idx[:5].tolist(), "count", idx.shape[0],
                      "nan", torch.isnan(outs[0]).sum().item())
    # fp16-residual epilogues (proj / fc2): in place, out of place, and with the LayerNorm statistics by-product
    for (m, n, k) in ((197 * 100, 768, 768), (197 * 300 + 5, 768, 3072), (197 * 880, 768, 768), (70001, 512, 128)):
        g = torch.Generator(device="cuda").manual_seed(m + n + 1)
        a = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
        w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).half()
        bias = torch.randn(n, device="cuda", generator=g)
        resid = (torch.randn(m, n, device="cuda", generator=g) + 3.0).half()
        nsl = n // 64
        res = []
        for L in (A, B):
            o1 = resid.clone()                                    # in place
            assert L.hcir_gemm_f16(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, 6, o1.data_ptr(), n, st) == 0
            o2 = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float16)   # out of place
            assert L.hcir_gemm_f16_resid(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, 6,
                                         resid.data_ptr(), o2.data_ptr(), n, st) == 0
            o3 = resid.clone()
            part = torch.full((nsl, m, 2), float("nan"), device="cuda")
            assert L.hcir_gemm_f16_fused(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, 6, o3.data_ptr(), n,
                                         None, None, part.data_ptr(), st) == 0
            res.append((o1, o2, o3, part))
        torch.cuda.synchronize()
        for nm, x, y in zip(("in place", "out of place", "with stats: rows", "with stats: slices"), res[0], res[1]):
            eq = torch.equal(x, y) and not torch.isnan(x.float()).any().item()
            bad += not eq
            print(f"m={m} n={n} k={k} resid {nm}: {'equal' if eq else 'DIFFERENT'}", flush=True)
        eq = torch.equal(res[0][0], res[0][1]) and torch.equal(res[0][0], res[0][2])
        bad += not eq
        print(f"m={m} n={n} k={k} resid forms agree with each other: {eq}", flush=True)
    print("RESULT", "all equal" if bad == 0 else f"{bad} differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
