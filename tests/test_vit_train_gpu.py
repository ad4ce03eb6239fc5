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


def test_forward_views_equals_separate_calls():
    """SHAM2.forward_views (one 3B-row backbone pass, the head per view) against three model(x) calls.  The backbone
    has no batch coupling and BatchNorm1d sees one view at a time either way, so the two differ only through the GEMM
    tile the row count selects (another fp32 summation order, so an fp16 output may round the other way: measured
    2e-3 on the outputs after twelve blocks and a 5-sample BatchNorm) - a wrong view split would be O(1)."""
    import copy
    from hcir.main_backbone import SHAM2
    torch.manual_seed(11)
    m1 = SHAM2("vit_b_16").cuda().train()
    m2 = copy.deepcopy(m1)
    views = [torch.randn(n, 3, 224, 224, device="cuda") for n in (5, 5, 5)]
    w = [torch.randn(5, 512, device="cuda") for _ in range(3)]
    o1 = [m1(v) for v in views]
    o2 = m2.forward_views(views)
    for a, b in zip(o1, o2):
        assert a.shape == b.shape and _rel(b, a) <= 1e-2
    sum((a * ww).sum() for a, ww in zip(o1, w)).backward()
    sum((a * ww).sum() for a, ww in zip(o2, w)).backward()
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), n1
        if p1.grad is not None:
            # sums over 15 samples of cancelling terms behind a 5-sample BatchNorm: 5e-2 measured on the class token's
            # gradient (1e-2 on the weights)
            # (+ an absolute floor: the final LayerNorm's bias gradient is the batch sum of a gradient that a
            # BatchNorm has just centred - zero in exact arithmetic, 2e-3 of rounding noise per element here)
            diff = (p2.grad.double() - p1.grad.double()).norm().item()
            assert diff <= 1e-1 * p1.grad.double().norm().item() + 5e-3 * p1.grad.numel() ** 0.5, (n1, diff)
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
