"""Experiment: one ViT-B/16 embedding forward (hcir engine, fp16 residual stream) captured in a HIP graph
(torch.cuda.CUDAGraph: hipStreamBeginCapture on the stream the C ABI launches into) against the same forward issued
launch by launch.   usage: python3 tools/exp_graph.py [batch=64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import vit_engine
from hcir.main_backbone import SHAM2

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
vit_engine.DEFAULT_RESID_DTYPE = torch.float16
torch.manual_seed(0)
model = SHAM2("vit_b_16").cuda().eval()
x = torch.randn(b, 3, 224, 224, device="cuda")


def fwd():
    return model.backbone.forward_cls(x, l2_normalize=True, want_f16=True)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = fwd()
    ref = ref[0] if isinstance(ref, tuple) else ref
    ref = ref.clone()
    eager = timeit(fwd)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
    out = out[0] if isinstance(out, tuple) else out
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    graph = timeit(g.replay)
print(f"batch {b}: eager {eager:.3f} ms  graph replay {graph:.3f} ms  ({eager / graph:.3f}x)  identical output: {same}")
