"""Config C3 (BASELINE.json configs[2]): one HSimCLR pretrain step on ONE MI355X - ViT-B/16 (SHAM2), NT-Xent over
the batch (1024 x 1024 cosine matrix at the config's batch), triplet + MSE, three differentiable backbone forwards
+ one momentum forward, backward, clip, Adam step.   usage: bench_train.py [batch=1024] [steps=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hair-centric-image-retrieval_amd"))
import torch
from hcir import vit_engine
from hcir.main_backbone import SHAM2
from hcir.pretrain_engine import SHAMTrainStep


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    vit_engine.DEFAULT_RESID_DTYPE = torch.float16      # the momentum forward runs on the inference engine
    torch.manual_seed(0)
    model = SHAM2("vit_b_16").cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    step = SHAMTrainStep(model, opt, scaler, temperature=0.5, warm_up_epochs=5)     # stage 1: random negatives
    gen = torch.Generator(device="cuda").manual_seed(1)
    batch = {"anchor": torch.randn(b, 3, 224, 224, device="cuda", generator=gen),
             "pos1": torch.randn(b, 3, 224, 224, device="cuda", generator=gen)}
    for _ in range(2):
        out = step(batch, epoch=0)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(batch, epoch=0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # ViT-B/16: 35.1 GFLOP per image forward; 3 differentiable forwards (x3 for fwd + bwd) + 1 momentum forward
    flops = b * 35.1e9 * (3 * 3 + 1)
    print(f"batch {b}: {dt*1e3:.1f} ms/step  {1/dt:.3f} steps/s  {b/dt:.0f} anchor-images/s  "
          f"{flops/dt/1e12:.0f} TFLOP/s (model flops)  peak HBM {torch.cuda.max_memory_allocated()/2**30:.1f} GiB  "
          f"loss {out['total']:.4f}", flush=True)
    # where the time goes: one differentiable forward, its backward, the momentum forward
    x = batch["anchor"]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    model.train()
    ev[0].record()
    cls = model.backbone.forward_cls(x)
    ev[1].record()
    cls.sum().backward()
    ev[2].record()
    with torch.no_grad():
        model.backbone_momentum.forward_cls(x)
    ev[3].record()
    torch.cuda.synchronize()
    print(f"  backbone forward (kept activations) {ev[0].elapsed_time(ev[1]):.1f} ms, backward {ev[1].elapsed_time(ev[2]):.1f} ms, "
          f"momentum forward (inference engine) {ev[2].elapsed_time(ev[3]):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
