"""GPU parity of the ViT backward and of the HSimCLR training step (config C3's
step, HP/src/pretrain_engine.py:618-751) against torch-CPU autograd through the oracle's functional ViT
(oracle/vit.py, fp32) — same state dict, same inputs.  Bar: gradients within 1e-2 relative."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ntxent as ont
from oracle import vit as ovit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(hcir_built):
    assert torch.cuda.is_available()


def _perturb(model, seed):
    """Non-trivial LayerNorm gains / biases and BatchNorm statistics (identity values would hide a dropped term)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and ("ln" in n or "norm" in n) and n.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
        model.backbone.cls_token.copy_(0.02 * torch.randn(model.backbone.cls_token.shape, generator=g))


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _oracle_cls(sd, x):
    """vitwrapper_forward with autograd ON (the oracle's functions are decorated with no_grad for inference)."""
    return ovit.vitwrapper_forward.__wrapped__(sd, x, "backbone.")[0]


def test_vit_backward_vs_oracle_autograd():
    from hcir.main_backbone import SHAM2
    torch.manual_seed(0)
    model = SHAM2("vit_b_16")
    _perturb(model, 1)
    b = 3
    x = torch.randn(b, 3, 224, 224)
    wgt = torch.randn(b, 768)
    # ---- oracle: fp32 autograd on the CPU over the same state dict
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.startswith("backbone_momentum"))
          for k, v in model.state_dict().items()}
    sd["backbone.pos_embedding"] = sd["backbone.encoder.pos_embedding"]      # one Parameter, two names (:536-537)
    sd["backbone.cls_token"] = sd["backbone.class_token"] if "backbone.class_token" in sd else sd["backbone.cls_token"]
    ref_cls = _oracle_cls(sd, x)
    (ref_cls * wgt).sum().backward()
    # ---- HIP: differentiable forward_cls
    model = model.cuda().train()
    cls = model.backbone.forward_cls(x.cuda())
    assert cls.requires_grad
    assert (1 - F.cosine_similarity(cls.detach().cpu().double(), ref_cls.detach().double(), dim=1)).max() <= 1e-3
    (cls * wgt.cuda()).sum().backward()
    named = dict(model.backbone.named_parameters())
    checked = 0
    worst = ("", 0.0)
    for name, p in named.items():
        key = "backbone." + name
        ref = sd[key].grad
        assert p.grad is not None and ref is not None, name
        err = _rel(p.grad.cpu(), ref)
        cos = F.cosine_similarity(p.grad.cpu().double().flatten(), ref.double().flatten(), dim=0).item()
        if err > worst[1]:
            worst = (name, err)
        assert err <= 1e-2 and cos >= 0.9999, (name, err, cos)
        checked += 1
    assert checked == len(named) and checked >= 12 * 12 + 6, checked
    print("worst relative gradient error:", worst)
    assert all(p.grad is None for p in model.backbone_momentum.parameters())
    # second backward through a fresh forward accumulates into .grad like torch
    g0 = named["encoder.layers.encoder_layer_5.mlp.0.weight"].grad.clone()
    (model.backbone.forward_cls(x.cuda()) * wgt.cuda()).sum().backward()
    assert _rel(named["encoder.layers.encoder_layer_5.mlp.0.weight"].grad, 2 * g0) <= 1e-6   # and deterministic


@pytest.mark.parametrize("scale", [1e-5, 3e4])
def test_vit_backward_is_scale_invariant(scale):
    """an incoming gradient of the size a batch-1024 mean loss produces WITHOUT a GradScaler (1e-5), or
    under a large loss scale, must give the same gradients as an O(1) one, times the scale: the backward renormalises
    d_cls by a power of two before its fp16 operands are formed (fp16 alone flushes dS = P (dP - D) to zero)."""
    from hcir.main_backbone import SHAM2
    torch.manual_seed(7)
    model = SHAM2("vit_b_16").cuda().train()
    x = torch.randn(2, 3, 224, 224, device="cuda")
    wgt = torch.randn(2, 768, device="cuda")
    names = ["encoder.layers.encoder_layer_0.self_attention.in_proj_weight", "encoder.layers.encoder_layer_0.ln_1.weight",
             "encoder.layers.encoder_layer_6.mlp.0.weight", "conv_proj.weight", "encoder.layers.encoder_layer_11.mlp.3.bias"]
    named = dict(model.backbone.named_parameters())

    def grads(s):
        model.zero_grad(set_to_none=True)
        (model.backbone.forward_cls(x) * (wgt * s)).sum().backward()
        return {n: named[n].grad.detach().clone() for n in names}

    g1, gs = grads(1.0), grads(scale)
    for n in names:
        assert torch.isfinite(gs[n]).all(), n
        assert _rel(gs[n] / scale, g1[n]) <= 5e-3, (n, _rel(gs[n] / scale, g1[n]))


@pytest.mark.parametrize("n", [5, 64])
def test_forward_views_equals_separate_calls(n):
    """SHAM2.forward_views (one 3n-row backbone pass, the head per view) against three model(x) calls, BOTH measured
    against a float64 ground truth: oracle/vit.py's functional ViTWrapper and a torch projection head, in double on the
    GPU, same parameters.  The fused pass may differ from the separate calls only through the GEMM tile its row count
    selects, so its error against float64 must stay within twice the separate calls' (+ a small floor), per output
    and per parameter gradient; with 64-sample views (a well-conditioned BatchNorm) both stay within 2e-2 on the outputs
    and 6e-2 on every gradient outright.
    n = 5 is the smoke case (a 5-sample BatchNorm amplifies any rounding); a wrong view split is O(1) in either."""
    import copy
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _fp64 import fp64_head
    from hcir.main_backbone import SHAM2
    torch.manual_seed(11)
    m1 = SHAM2("vit_b_16").cuda().train()
    _perturb(m1, 4)
    m2 = copy.deepcopy(m1)
    views = [torch.randn(n, 3, 224, 224, device="cuda") for _ in range(3)]
    w = [torch.randn(n, 512, device="cuda") for _ in range(3)]
    o1 = [m1(v) for v in views]
    o2 = m2.forward_views(views)
    sum((a * ww).sum() for a, ww in zip(o1, w)).backward()
    sum((a * ww).sum() for a, ww in zip(o2, w)).backward()
    # ---- float64 ground truth (same state dict; the head sees one view at a time, in order)
    sd = {k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point and k.startswith("backbone."))
          for k, v in m1.state_dict().items() if not k.startswith(("backbone_momentum", "projection_head"))}
    sd["backbone.pos_embedding"] = sd["backbone.encoder.pos_embedding"]      # one Parameter, two names (:536-537)
    sd["backbone.cls_token"] = sd["backbone.class_token"] if "backbone.class_token" in sd else sd["backbone.cls_token"]
    head = fp64_head(m1.projection_head, "cuda")
    o64 = [head(_oracle_cls(sd, v.double())) for v in views]
    sum((a * ww.double()).sum() for a, ww in zip(o64, w)).backward()
    g64 = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    for (hn, hp), tp in zip(m1.projection_head.named_parameters(), head.parameters()):
        g64["projection_head." + hn] = tp.grad
    worst, checked = 0.0, 0
    for a, b, t in zip(o1, o2, o64):
        e_s, e_v = _rel(a, t), _rel(b, t)
        assert e_v <= 2.0 * e_s + 1e-3, ("output", e_v, e_s)
        if n >= 64:
            assert e_s <= 2e-2 and e_v <= 2e-2
    # error norms are taken against the gradient's own norm plus a floor of 1e-3 of the LARGEST per-element gradient
    # RMS of the model: a gradient that is zero in exact arithmetic (the final LayerNorm's bias behind a centring
    # BatchNorm, the last block's key bias) has only rounding noise to compare
    rms = max((t.norm() / t.numel() ** 0.5).item() for t in g64.values())
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), n1
        if p1.grad is None or n1 not in g64:
            continue
        t = g64[n1].reshape(p1.grad.shape)
        floor = 1e-3 * rms * t.numel() ** 0.5
        e_s = (p1.grad.double() - t).norm().item() / (t.norm().item() + floor)
        e_v = (p2.grad.double() - t).norm().item() / (t.norm().item() + floor)
        worst = max(worst, e_v / max(e_s, 1e-12))
        checked += 1
        assert e_v <= 2.0 * e_s + 2e-3, (n1, e_v, e_s)
        if n >= 64:
            # both paths against float64 outright (measured: 4.4e-2 on the class token's gradient, a sum over 192
            # samples of nearly cancelling terms, identical for the two paths; <= 1e-2 on the weights)
            bar = 1e-1 if n1.startswith("projection_head.layers.1.") or n1.startswith("projection_head.layers.4.") else 6e-2
            assert e_v <= bar and e_s <= bar, (n1, e_v, e_s)   # (7.0e-2 on the first BatchNorm's bias, both paths)
    assert checked > 100
    print(f"n={n}: worst fused / separate error ratio {worst:.2f} over {checked} gradients")
    for (n1, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        if "num_batches" in n1:
            assert int(b1) == int(b2), n1
        else:
            assert _rel(b2, b1) <= 1e-2, n1


@pytest.mark.parametrize("kwargs", [dict(keep_recomputable=False), dict(cls_only_last=False),
                                    dict(keep_recomputable=False, cls_only_last=False)])
def test_trainer_variants_match_default(kwargs):
    """`keep_recomputable=False` (LayerNorm outputs and gelu(u) recomputed in the backward instead of kept: 11.8 instead
    of 15.4 KB per token and block) and `cls_only_last=False` (the last block on every token) against the default
    trainer: same class-token output, every gradient within the fp16 rounding the different operand paths leave
    (gelu recomputed from the fp16 pre-activation instead of the fp32 accumulator: one fp16 ulp on h)."""
    import copy
    from hcir.main_backbone import SHAM2
    from hcir.vit_train import VitTrainer
    torch.manual_seed(21)
    m1 = SHAM2("vit_b_16").cuda().train()
    _perturb(m1, 3)
    m2 = copy.deepcopy(m1)
    dev = torch.device("cuda")
    m2.backbone._trainer = VitTrainer(m2.backbone._spec(), dev, **kwargs)
    x = torch.randn(3, 3, 224, 224, device="cuda")
    wgt = torch.randn(3, 768, device="cuda")
    c1 = m1.backbone.forward_cls(x)
    c2 = m2.backbone.forward_cls(x)
    assert _rel(c2.detach(), c1.detach()) <= 2e-3
    (c1 * wgt).sum().backward()
    (c2 * wgt).sum().backward()
    for (n, p1), (_, p2) in zip(m1.backbone.named_parameters(), m2.backbone.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), n
        if p1.grad is not None:
            assert _rel(p2.grad, p1.grad) <= 1e-2, (n, _rel(p2.grad, p1.grad))


def test_vit_backward_gradient_direction_decreases_loss():
    """Independent of any oracle: a small step against the HIP gradient lowers the loss by ~ lr |g|^2."""
    from hcir.main_backbone import SHAM2
    torch.manual_seed(2)
    model = SHAM2("vit_b_16").cuda().train()
    x = torch.randn(4, 3, 224, 224, device="cuda")
    tgt = torch.randn(4, 768, device="cuda")
    loss = lambda: ((model.backbone.forward_cls(x) - tgt) ** 2).mean()
    l0 = loss()
    l0.backward()
    params = [p for p in model.backbone.parameters() if p.grad is not None]
    g2 = sum(float((p.grad.double() ** 2).sum()) for p in params)
    lr = 1e-3 / max(g2 ** 0.5, 1e-12)
    with torch.no_grad():
        for p in params:
            p.add_(p.grad, alpha=-lr)
        l1 = loss()
    pred = lr * g2
    assert float(l0) - float(l1) >= 0.5 * pred, (float(l0), float(l1), pred)


def test_sham_train_step_matches_oracle_forward_and_trains():
    """One HSimCLR step (HP/src/pretrain_engine.py:618-751): loss terms against a CPU restatement (oracle ViT +
    torch head modules in train mode + torch's TripletMarginLoss / mse_loss + the oracle's NT-Xent), then the
    optimiser step must move the parameters and the momentum twins, and a second step must run."""
    from hcir.main_backbone import SHAM2
    from hcir.pretrain_engine import SHAMTrainStep
    torch.manual_seed(3)
    model = SHAM2("vit_b_16")
    _perturb(model, 4)
    b = 8
    g = torch.Generator().manual_seed(5)
    batch = {"anchor": torch.randn(b, 3, 224, 224, generator=g), "pos1": torch.randn(b, 3, 224, 224, generator=g)}
    neg_idx = torch.tensor([(i + 3) % b for i in range(b)])
    # ---- CPU restatement of the forward half of the step (momentum update included: m = 0.99)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in list(sd):
        if k.startswith("backbone_momentum.") or k.startswith("projection_head_momentum."):
            src = k.replace("backbone_momentum.", "backbone.").replace("projection_head_momentum.", "projection_head.")
            if sd[k].dtype.is_floating_point and "running" not in k:
                sd[k] = sd[k] * 0.99 + sd[src] * (1.0 - 0.99)
    head, head_m = copy.deepcopy(model.projection_head).train(), copy.deepcopy(model.projection_head_momentum).train()
    with torch.no_grad():
        for pm, p in zip(head_m.parameters(), head.parameters()):
            pm.copy_(pm * 0.99 + p * (1.0 - 0.99))
    # the step's DEFAULT path applies positive_transform (HP/src/pretrain_engine.py:686): torchvision draws one angle
    # and one sigma per call from torch's CPU generator; the same seed gives the HIP step the same two numbers
    from oracle import transform as otf
    torch.manual_seed(77)
    angle = float(torch.empty(1).uniform_(-15.0, 15.0).item())
    sigma = float(torch.empty(1).uniform_(0.1, 0.5).item())
    with torch.no_grad():
        z_neg = head(ovit.vitwrapper_forward(sd, batch["pos1"][neg_idx], "backbone.")[0])
        z_pos = head(ovit.vitwrapper_forward(sd, otf.positive_transform(batch["pos1"], angle, sigma), "backbone.")[0])
        z_anc = head(ovit.vitwrapper_forward(sd, batch["anchor"], "backbone.")[0])
    n_neg, n_pos, n_anc = (F.normalize(z, dim=1) for z in (z_neg, z_pos, z_anc))
    ref_trip = float(torch.nn.TripletMarginLoss(margin=0.5, p=2, eps=1e-7)(n_anc, n_pos, n_neg))
    ref_con = float(ont.ntxent_lightly(n_pos, n_anc, 0.5))
    # ---- HIP step
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    step = SHAMTrainStep(model, opt, scaler, temperature=0.5, momentum=0.99, warm_up_epochs=0,
                         ablation="No masked positive")
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    torch.manual_seed(77)           # the two uniform_ draws of positive_transform are the first CPU draws of the step
    out = step(dev_batch, epoch=0, negative_idx=neg_idx.cuda())
    assert abs(out["triplet"] - ref_trip) <= 5e-3 * max(1.0, abs(ref_trip)), (out["triplet"], ref_trip)
    assert abs(out["contrastive"] - ref_con) <= 5e-3 * max(1.0, abs(ref_con)), (out["contrastive"], ref_con)
    assert np.isfinite(out["total"]) and out["mse"] >= 0.0
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters() if p.requires_grad)
    assert moved >= 150, moved                                    # Adam moved (nearly) every trainable tensor
    # the momentum twins moved by (1 - m) * (online - ema) BEFORE the optimiser step
    k = "backbone_momentum.encoder.layers.encoder_layer_0.mlp.0.weight"
    want = before[k] * 0.99 + before[k.replace("backbone_momentum", "backbone")] * (1.0 - 0.99)
    assert torch.allclose(dict(model.named_parameters())[k], want, rtol=0, atol=1e-7)
    out2 = step(dev_batch, epoch=0, negative_idx=neg_idx.cuda())
    assert np.isfinite(out2["total"])
    # stage-1 path: random negatives + masked positives through hcir_positive_masking
    step1 = SHAMTrainStep(model, opt, scaler, warm_up_epochs=5)
    out3 = step1(dev_batch, epoch=0)
    assert np.isfinite(out3["total"]) and out3["mse"] > 0.0
