"""GPU parity of the training kernels against torch-CPU autograd in float64 on the
same (fp16-rounded) inputs: GELU fwd/bwd, LayerNorm backward, bias column sums, the weight-gradient (TN) GEMM,
attention backward, TripletMarginLoss and mse_loss forward/backward (HP/src/pretrain_engine.py:96-97,717-745).
Tolerances are stated per test: fp16 outputs carry 2^-11 relative rounding, fp32 outputs fp32 summation error."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(hcir_built):
    assert torch.cuda.is_available()


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def test_gelu_fwd_bwd():
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(0)
    u = (torch.randn(1000, 3072, generator=g) * 2.0).half()
    u[0, :8] = torch.tensor([0.0, -0.0, 8.0, -8.0, 1e-4, -1e-4, 30.0, -30.0]).half()
    dh = torch.randn(1000, 3072, generator=g).half()
    ud = u.double().requires_grad_(True)
    ref = F.gelu(ud)
    ref.backward(dh.double())
    h = T.gelu_fwd(u.cuda()).cpu()
    du = T.gelu_bwd(u.cuda(), dh.cuda()).cpu()
    assert (h.double() - ref.detach()).abs().max() <= 2e-3 * 8 and _rel(h, ref.detach()) <= 1e-3
    assert (du.double() - ud.grad).abs().max() <= 4e-3          # |dh| <= ~5, fp16 output rounding
    assert torch.isfinite(h).all() and torch.isfinite(du).all()


@pytest.mark.parametrize("xdt", [torch.float16, torch.float32])
@pytest.mark.parametrize("rows,d", [(197 * 5, 768), (64, 1024), (7, 128), (3000, 768)])
def test_layernorm_bwd(xdt, rows, d):
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(rows, d, generator=g) * 1.5 + 0.3).to(xdt)
    dy = torch.randn(rows, d, generator=g).half()
    gamma = 1.0 + 0.2 * torch.randn(d, generator=g)
    dres = torch.randn(rows, d, generator=g)
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = torch.zeros(d, dtype=torch.float64, requires_grad=True)
    y = F.layer_norm(xd, (d,), gd, bd, 1e-6)
    y.backward(dy.double())
    out = torch.empty(rows, d, device="cuda")
    dgam = torch.full((d,), 0.5, device="cuda")
    dbet = torch.full((d,), -0.25, device="cuda")
    T.layernorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), 1e-6, dres.cuda(), out, dgam, dbet, accumulate=True)
    assert _rel(out.cpu(), xd.grad + dres.double()) <= 2e-5
    assert _rel(dgam.cpu() - 0.5, gd.grad) <= 2e-5
    assert _rel(dbet.cpu() + 0.25, bd.grad) <= 2e-5
    out2 = torch.empty(rows, d, device="cuda")
    T.layernorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), 1e-6, None, out2, dgam, dbet, accumulate=False)
    assert _rel(out2.cpu(), xd.grad) <= 2e-5 and _rel(dgam.cpu(), gd.grad) <= 2e-5
    T.layernorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), 1e-6, None, out, dgam, dbet, accumulate=False)
    assert torch.equal(out, out2)           # deterministic (no atomics)


def test_layernorm_bwd_strided_cls_rows():
    """The final LayerNorm sees only the class-token rows: row pitch T * D for x and the residual gradient."""
    from hcir import train_ops as T
    b, t, d = 6, 197, 768
    g = torch.Generator().manual_seed(2)
    tok = torch.randn(b, t, d, generator=g).half()
    dy = torch.randn(b, d, generator=g).half()
    gamma = 1.0 + 0.1 * torch.randn(d, generator=g)
    xd = tok[:, 0].double().requires_grad_(True)
    F.layer_norm(xd, (d,), gamma.double(), torch.zeros(d, dtype=torch.float64), 1e-6).backward(dy.double())
    dres = torch.zeros(b, t, d, device="cuda")
    dgam, dbet = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    T.layernorm_bwd(tok.cuda(), dy.cuda(), gamma.cuda(), 1e-6, None, dres, dgam, dbet, accumulate=False, rows=b,
                    ldx=t * d, ldr=t * d)
    assert _rel(dres[:, 0].cpu(), xd.grad) <= 2e-5
    assert float(dres[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,d", [(197 * 5, 768), (3000, 768), (7, 128)])
def test_layernorm_bwd_fused_copy_and_colsum(rows, d):
    """hcir_layernorm_bwd_fused: same dres_out / dgamma / dbeta as hcir_layernorm_bwd (bit for bit), plus the fp16 copy
    of dres_out (== hcir_add_f32_f16 of it, bit for bit) and that copy's column sums (vs float64)."""
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(rows, d, generator=g) * 1.5 + 0.3).half().cuda()
    dy = torch.randn(rows, d, generator=g).half().cuda()
    gamma = (1.0 + 0.2 * torch.randn(d, generator=g)).cuda()
    dres = torch.randn(rows, d, generator=g).cuda()
    o0, o1 = torch.empty(rows, d, device="cuda"), torch.empty(rows, d, device="cuda")
    ga0, be0, ga1, be1 = (torch.zeros(d, device="cuda") for _ in range(4))
    T.layernorm_bwd(x, dy, gamma, 1e-6, dres, o0, ga0, be0, accumulate=False)
    h16 = torch.full((rows, d), 7.0, dtype=torch.float16, device="cuda")
    cs = torch.full((d,), 3.0, device="cuda")
    T.layernorm_bwd(x, dy, gamma, 1e-6, dres, o1, ga1, be1, accumulate=False, dres16=h16, dres_colsum=cs)
    assert torch.equal(o0, o1) and torch.equal(ga0, ga1) and torch.equal(be0, be1)
    assert torch.equal(h16, T.add_to_f16(o0))
    assert (cs.cpu().double() - h16.cpu().double().sum(0)).abs().max() <= 1e-4 * np.sqrt(rows)
    # the copy alone (no column sums)
    h2 = torch.zeros_like(h16)
    T.layernorm_bwd(x, dy, gamma, 1e-6, dres, o1, ga1, be1, accumulate=False, dres16=h2)
    assert torch.equal(h2, h16)


@pytest.mark.parametrize("m,n", [(1970, 3072), (50432, 3072), (13, 8)])
def test_gelu_bwd_colsum_equals_the_two_kernels(m, n):
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(12)
    u = (torch.randn(m, n, generator=g) * 2.0).half().cuda()
    dh = torch.randn(m, n, generator=g).half().cuda()
    du_ref = T.gelu_bwd(u, dh)
    cs_ref = torch.zeros(n, device="cuda")
    T.colsum(du_ref, cs_ref, accumulate=False)
    du = torch.empty_like(u)
    cs = torch.full((n,), 5.0, device="cuda")
    T.gelu_bwd_colsum(u, dh, du, cs)
    assert torch.equal(du, du_ref) and torch.equal(cs, cs_ref)
    dh2 = dh.clone()
    T.gelu_bwd_colsum(u, dh2, dh2, cs, accumulate=True)            # in place, accumulating
    assert torch.equal(dh2, du_ref) and torch.allclose(cs, 2 * cs_ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("m,n", [(50432, 768), (1000, 2304), (13, 8), (4096, 3072)])
def test_colsum(m, n):
    from hcir import train_ops as T
    x = torch.randn(m, n, generator=torch.Generator().manual_seed(3)).half()
    out = torch.full((n,), 2.0, device="cuda")
    T.colsum(x.cuda(), out, accumulate=True)
    ref = x.double().sum(0) + 2.0
    assert (out.cpu().double() - ref).abs().max() <= 1e-4 * np.sqrt(m)
    T.colsum(x.cuda(), out, accumulate=False)
    assert (out.cpu().double() - (ref - 2.0)).abs().max() <= 1e-4 * np.sqrt(m)


@pytest.mark.parametrize("m,n,k", [(64, 256, 256), (1984, 768, 768), (12608, 2304, 768), (6400, 768, 3072),
                                   (50432, 768, 768)])
def test_gemm_tn(m, n, k):
    """dW = A^T B with M the slow index of both operands (transposed LDS reads), split over workgroups."""
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(4)
    a = (torch.randn(m, n, generator=g) * 0.5).half()
    b = (torch.randn(m, k, generator=g) * 0.5).half()
    # asymmetric structure: a swapped row/column map cannot pass (cdna_hip_programming.md §3)
    a[:, 0] = 1.0
    a[:, 1] = torch.arange(m).half() % 7
    b[:, 3] = torch.arange(m).half() % 5
    dw = torch.full((n, k), 1.0, device="cuda")
    T.gemm_tn(a.cuda(), b.cuda(), dw, accumulate=True)
    ref = a.double().t() @ b.double() + 1.0
    err = (dw.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * m * 0.25 + 1e-3, err          # fp32 accumulation of exact fp16 products
    dw2 = torch.empty((n, k), device="cuda")
    T.gemm_tn(a.cuda(), b.cuda(), dw2, accumulate=False)
    T.gemm_tn(a.cuda(), b.cuda(), dw, accumulate=False)
    assert torch.equal(dw, dw2)                          # deterministic split reduction
    from hcir import HcirError
    with pytest.raises(HcirError):
        T.gemm_tn(a[:63].cuda(), b[:63].cuda(), dw)      # M % 64


def _attn_ref(qkv, b, t, heads, scale):
    q, k, v = qkv.reshape(b, t, 3, heads, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q * scale) @ k.transpose(-2, -1), dim=-1)
    return (p @ v).transpose(1, 2).reshape(b, t, heads * 64)


@pytest.mark.parametrize("b,t,heads", [(2, 197, 12), (1, 33, 2), (3, 256, 4), (1, 64, 1), (2, 17, 3)])
def test_attention_backward(b, t, heads):
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(b, t, 3 * heads * 64, generator=g) * 0.8).half()
    qkv[0, 3, :64] *= 4.0                                  # a spiked query row: sharp softmax
    dout = torch.randn(b, t, heads * 64, generator=g).half()
    scale = 64 ** -0.5
    qd = qkv.double().requires_grad_(True)
    ref_out = _attn_ref(qd, b, t, heads, scale)
    ref_out.backward(dout.double())
    out = torch.empty(b, t, heads * 64, dtype=torch.float16, device="cuda")
    lse = torch.empty(b, heads, t, dtype=torch.float32, device="cuda")
    T.attn_fwd_lse(qkv.cuda(), b, t, heads, scale, out, lse)
    assert (out.cpu().double() - ref_out.detach()).abs().max() <= 4e-3
    # lse against the definition (log2 domain)
    q, k, _ = qkv.double().reshape(b, t, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref_lse = torch.logsumexp((q * scale) @ k.transpose(-2, -1), dim=-1) / np.log(2.0)
    assert (lse.cpu().double() - ref_lse).abs().max() <= 1e-3
    dqkv = torch.full((b, t, 3 * heads * 64), float("nan"), dtype=torch.float16, device="cuda")
    T.attn_bwd(qkv.cuda(), out, dout.cuda(), lse, b, t, heads, scale, dqkv)
    got = dqkv.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - qd.grad).abs().max().item()
    assert err <= 1.5e-2 * qd.grad.abs().max().item() + 2e-3, (err, qd.grad.abs().max().item())
    # per section (dq, dk, dv) relative agreement
    for s0 in range(3):
        sl = slice(s0 * heads * 64, (s0 + 1) * heads * 64)
        assert _rel(got[..., sl], qd.grad[..., sl]) <= 2e-2


@pytest.mark.parametrize("b,t,heads", [(26, 197, 12), (48, 197, 12), (30, 193, 12), (70, 65, 8)])
def test_attention_backward_persistent_loop(b, t, heads):
    """More (image, head) items than workgroups (the grid is min(items, 256)): the loop-carried paths of the persistent
    kernel run — the counted wait on the next item's fragments, the transfers issued under pass 2, the dQ stores
    deferred behind the next barrier.  312, 576, 360 (t = 193: a last wave with ONE live row) and 560 items.
    Checked against the fp64 definition, and bit for bit against the same kernel fed ONE image at a time (12 or 8
    items: one item per workgroup, nothing carried from item to item)."""
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(50 + b)
    qkv = (torch.randn(b, t, 3 * heads * 64, generator=g) * 0.8).half()
    qkv[b // 2, 3, :64] *= 4.0
    dout = torch.randn(b, t, heads * 64, generator=g).half()
    scale = 64 ** -0.5
    out = torch.empty(b, t, heads * 64, dtype=torch.float16, device="cuda")
    lse = torch.empty(b, heads, t, dtype=torch.float32, device="cuda")
    qc, dc = qkv.cuda(), dout.cuda()
    T.attn_fwd_lse(qc, b, t, heads, scale, out, lse)
    dqkv = torch.full((b, t, 3 * heads * 64), float("nan"), dtype=torch.float16, device="cuda")
    T.attn_bwd(qc, out, dc, lse, b, t, heads, scale, dqkv)
    assert torch.isfinite(dqkv).all()
    # fp64 definition, a few images at a time (the 48-image case would hold 48 x 12 x 197 x 197 doubles four times)
    for i0 in range(0, b, 8):
        sl = slice(i0, min(i0 + 8, b))
        qd = qkv[sl].double().requires_grad_(True)
        ref_out = _attn_ref(qd, qd.shape[0], t, heads, scale)
        ref_out.backward(dout[sl].double())
        assert (out[sl].cpu().double() - ref_out.detach()).abs().max() <= 4e-3
        got = dqkv[sl].cpu().double()
        err = (got - qd.grad).abs().max().item()
        assert err <= 1.5e-2 * qd.grad.abs().max().item() + 2e-3, (i0, err)
        for s0 in range(3):
            c = slice(s0 * heads * 64, (s0 + 1) * heads * 64)
            assert _rel(got[..., c], qd.grad[..., c]) <= 2e-2
    # one image at a time: every workgroup handles exactly one item
    for i in sorted({0, 1, 256 // heads, 256 // heads + 1, b // 2, b - 1}):
        o1 = torch.empty(1, t, heads * 64, dtype=torch.float16, device="cuda")
        l1 = torch.empty(1, heads, t, dtype=torch.float32, device="cuda")
        T.attn_fwd_lse(qc[i:i + 1].contiguous(), 1, t, heads, scale, o1, l1)
        assert torch.equal(o1[0], out[i]) and torch.equal(l1[0], lse[i])
        d1 = torch.full((1, t, 3 * heads * 64), float("nan"), dtype=torch.float16, device="cuda")
        T.attn_bwd(qc[i:i + 1].contiguous(), o1, dc[i:i + 1].contiguous(), l1, 1, t, heads, scale, d1)
        assert torch.equal(d1[0], dqkv[i]), f"image {i}: the persistent loop changes the result"


@pytest.mark.parametrize("b,t,heads", [(26, 197, 12), (50, 193, 12)])
def test_attention_cls_backward_persistent_loop(b, t, heads):
    """hcir_attn_cls_fwd_lse / hcir_attn_cls_bwd (the last block's attention for the CLS row alone) with more items
    than workgroups: against the fp64 definition and bit for bit against one image at a time."""
    from hcir import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(90 + b)
    qkv = (torch.randn(b, t, 3 * heads * 64, generator=g) * 0.8).half()
    dout = torch.randn(b, heads * 64, generator=g).half()
    scale = 64 ** -0.5

    def run(q, do, n):
        o = torch.empty(n, heads * 64, dtype=torch.float16, device="cuda")
        l = torch.empty(n, heads, dtype=torch.float32, device="cuda")
        _lib.check(L.hcir_attn_cls_fwd_lse(q.data_ptr(), n, t, heads, 64, scale, o.data_ptr(), l.data_ptr(), st), "fwd")
        d = torch.full((n, t, 3 * heads * 64), float("nan"), dtype=torch.float16, device="cuda")
        _lib.check(L.hcir_attn_cls_bwd(q.data_ptr(), o.data_ptr(), do.data_ptr(), l.data_ptr(), n, t, heads, 64, scale,
                                       d.data_ptr(), st), "bwd")
        return o, l, d
    qc, dc = qkv.cuda(), dout.cuda()
    out, lse, dqkv = run(qc, dc, b)
    assert torch.isfinite(dqkv).all()
    qd = qkv.double().requires_grad_(True)
    ref = _attn_ref(qd, b, t, heads, scale)[:, 0]
    ref.backward(dout.double())
    assert (out.cpu().double() - ref.detach()).abs().max() <= 4e-3
    got = dqkv.cpu().double()
    err = (got - qd.grad).abs().max().item()
    assert err <= 1.5e-2 * qd.grad.abs().max().item() + 2e-3, err
    for i in sorted({0, 256 // heads, 256 // heads + 1, b - 1}):
        o1, l1, d1 = run(qc[i:i + 1].contiguous(), dc[i:i + 1].contiguous(), 1)
        assert torch.equal(o1[0], out[i]) and torch.equal(l1[0], lse[i]) and torch.equal(d1[0], dqkv[i])


def test_triplet_and_mse_losses():
    from hcir import train_ops as T
    g = torch.Generator().manual_seed(6)
    b, d = 256, 512
    a, p, n = (F.normalize(torch.randn(b, d, generator=g), dim=1) for _ in range(3))
    p = F.normalize(a + 0.3 * p, dim=1)                      # positives close: some rows inactive at margin 0.5
    for margin in (0.7, 0.5, 0.05):
        ad, pd, nd = (t.clone().requires_grad_(True) for t in (a, p, n))
        ref = torch.nn.TripletMarginLoss(margin=margin, p=2, eps=1e-7)(ad, pd, nd)
        (ref * 3.0).backward()
        ac, pc, nc = (t.cuda().requires_grad_(True) for t in (a, p, n))
        got = T.TripletMarginLoss(margin=margin, p=2, eps=1e-7)(ac, pc, nc)
        (got * 3.0).backward()
        assert abs(float(got) - float(ref)) <= 1e-6
        for x, y in ((ac, ad), (pc, pd), (nc, nd)):
            assert (x.grad.cpu() - y.grad).abs().max() <= 1e-7
    xd, yd = a.clone().requires_grad_(True), p.clone().requires_grad_(True)
    ref = F.mse_loss(xd, yd)
    ref.backward()
    xc, yc = a.cuda().requires_grad_(True), p.cuda().requires_grad_(True)
    got = T.mse_loss(xc, yc)
    got.backward()
    assert abs(float(got) - float(ref)) <= 1e-9 + 1e-6 * float(ref)
    assert (xc.grad.cpu() - xd.grad).abs().max() <= 1e-10 and (yc.grad.cpu() - yd.grad).abs().max() <= 1e-10
    with pytest.raises(NotImplementedError):
        T.TripletMarginLoss(margin=1.0, p=1)
