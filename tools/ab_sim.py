"""A/B of libhcir builds on hcir_sim_topk, interleaved rounds in ONE process (cdna_hip_programming.md §5.4 rule 24).
usage: python3 tools/ab_sim.py tag=path [tag=path ...]     ('base=' = the in-tree library)
Cases: 1 M x 768 fp16 (k = 16; 1 / 32 / 64 / 128 / 880 queries), 1.25 M x 1024 fp16 top-50 (32 / 64 queries)."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import _lib


def load(path):
    l = ctypes.CDLL(path)
    for name in ("hcir_sim_topk_workspace_bytes", "hcir_sim_topk"):
        fn = getattr(l, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    return l


def main():
    libs = []
    for spec in sys.argv[1:]:
        tag, _, path = spec.partition("=")
        libs.append((tag, load(path or _lib.LIB_PATH)))
    st = torch.cuda.current_stream().cuda_stream
    for (ng, d, k, nqs) in ((1_000_000, 768, 16, (1, 32, 64, 128, 880)), (1_250_000, 1024, 50, (32, 64))):
        g = torch.empty(ng, d, device="cuda", dtype=torch.float16)
        for s in range(0, ng, 250_000):
            g[s:s + 250_000] = torch.nn.functional.normalize(torch.randn(min(250_000, ng - s), d, device="cuda"), dim=1).half()
        for nq in nqs:
            q = torch.nn.functional.normalize(torch.randn(nq, d, device="cuda"), dim=1).half()
            val = torch.empty(nq, k, device="cuda")
            idx = torch.empty(nq, k, dtype=torch.int64, device="cuda")
            ws = {t: torch.empty(L.hcir_sim_topk_workspace_bytes(nq, ng, d, k, 1), dtype=torch.uint8, device="cuda") for t, L in libs}
            call = lambda t, L: L.hcir_sim_topk(q.data_ptr(), nq, g.data_ptr(), ng, d, k, 1, None, None, 0, val.data_ptr(),
                                                idx.data_ptr(), ws[t].data_ptr(), ws[t].numel(), st)
            ref = None
            for t, L in libs:
                for _ in range(20):
                    assert call(t, L) == 0
                torch.cuda.synchronize()
                if ref is None:
                    ref = (val.clone(), idx.clone())
                else:
                    assert torch.equal(ref[1], idx) and torch.equal(ref[0], val), f"{t}: results differ"
            times = {t: [] for t, _ in libs}
            for r in range(7):
                for t, L in libs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        call(t, L)
                    e1.record()
                    torch.cuda.synchronize()
                    times[t].append(e0.elapsed_time(e1) / 20 * 1e3)
            gb = ng * d * 2 / 1e3
            print(f"ng={ng} d={d} k={k} nq={nq:4d}: " + "  ".join(
                f"{t} {statistics.median(times[t]):7.1f} us ({gb / statistics.median(times[t]):5.0f} GB/s)" for t, _ in libs), flush=True)
        del g


if __name__ == "__main__":
    main()
