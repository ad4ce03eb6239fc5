"""Config C5 slice on ONE MI355X (BASELINE.json configs[4]): ViT-L/14 fp16 embed + top-50 over this GPU's
1.25 M x 1024 fp16 shard of a 10 M gallery (8-GPU row sharding).  Prints img/s of the embed, ms of the scan."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import ops, vit_engine
from hcir.models_vit import vit_large_patch14

def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    vit_engine.DEFAULT_RESID_DTYPE = torch.float16
    torch.manual_seed(0)
    m = vit_large_patch14(drop_path_rate=0.0, global_pool=True, init_values=None).eval().cuda()
    x = torch.randn(b, 3, 224, 224, device="cuda")
    ng, d, k = 1_250_000, 1024, 50
    g = torch.empty(ng, d, device="cuda", dtype=torch.float16)
    for s in range(0, ng, 250_000):
        t = torch.randn(250_000, d, device="cuda")
        g[s:s + 250_000] = (t / t.norm(dim=1, keepdim=True)).half()
    with torch.no_grad():
        for _ in range(2):
            f = m.forward_features(x)[:, 0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            f = m.forward_features(x)[:, 0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        flops = 162e9 * b
        print(f"ViT-L/14 embed batch {b}: {dt*1e3:.2f} ms  {b/dt:.0f} img/s  {flops/dt/1e12:.0f} TFLOP/s", flush=True)
        q = torch.nn.functional.normalize(f.float(), dim=1).half()
        for nq in (32, 64, b):
            qq = q[:nq].contiguous()
            ops.sim_topk(qq, g, k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.sim_topk(qq, g, k)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"top-{k} of {nq} queries over {ng} x {d} fp16: {ms:.3f} ms  {ng*d*2/ms/1e6:.0f} GB/s", flush=True)

if __name__ == "__main__":
    main()
