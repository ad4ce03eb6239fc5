#!/usr/bin/env python3
"""A few launches of each ViT-B/16 GEMM shape at batch 880 on the library HCIR_LIB_PATH names: the process
`rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) is run over, one build variant at a time (tools/pmc_gemm_variants.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 880
    L = _lib.lib()
    m = b * 197
    st = torch.cuda.current_stream().cuda_stream
    for (n, k, epi) in ((2304, 768, 0), (768, 768, 6), (3072, 768, 1), (768, 3072, 6)):
        a = (torch.randn(m, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
        bias = torch.randn(n, device="cuda")
        out = torch.zeros(m, n, device="cuda", dtype=torch.float16)
        resid = torch.zeros(m, n, device="cuda", dtype=torch.float16) if epi == 6 else None
        for _ in range(4):
            if epi == 6:
                rc = L.hcir_gemm_f16_resid(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, epi,
                                           resid.data_ptr(), out.data_ptr(), n, st)
            else:
                rc = L.hcir_gemm_f16(a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), None, m, n, k, epi, out.data_ptr(), n, st)
            assert rc == 0, rc
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
