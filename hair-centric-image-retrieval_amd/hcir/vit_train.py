"""hcir.vit_train — training-mode forward and backward of the ViT on MI355X through libhcir
(the backbone half of `self.model(x)` + `.backward()` in
HP/src/pretrain_engine.py:682-745).

`VitTrainer.forward(x)` runs
    patch_embed -> depth x [ LN1 -> qkv GEMM -> attention (+ log-sum-exp) -> proj GEMM + residual ->
                             LN2 -> fc1 GEMM -> GELU -> fc2 GEMM + residual ] -> final LN of the class token
keeping, per block, what the backward needs (fp16: block input, qkv, attention output, the residual after the
attention, the fc1 pre-activation, and — by default, `keep_recomputable=True` — both LayerNorm outputs and the GELU
output, which are the A operands of the weight-gradient GEMMs; fp32: the attention's row log-sum-exp): 15.4 KB per
token and block, ~165 GB for config C3's three 1024-image forwards of the 288 GB.  `keep_recomputable=False`
recomputes the LayerNorm / GELU outputs in the backward instead (three HBM-bound launches per block, 11.8 KB per
token and block).  `VitTrainer.backward(saved, d_cls)` walks the blocks in reverse:
    dgrad    dX = dY . W          hcir_gemm_f16 against a transposed fp16 copy of the weight
    wgrad    dW = dY^T . X        hcir_gemm_f16_tn (operands as stored, transposed LDS reads, split-M, deterministic)
    bias     db = colsum(dY)      hcir_colsum_f16
    GELU', LayerNorm', attention' hcir_gelu_bwd_f16, hcir_layernorm_bwd, hcir_attn_bwd
The residual gradient travels in fp32, GEMM operands in fp16 (a GradScaler's loss scale passes through linearly,
as under the reference's fp16 autocast).  `vit_cls_with_grad` wraps both in a torch.autograd.Function so that
SHAM2.forward composes with torch's optimizer, GradScaler and clip_grad_norm_ exactly as in the reference loop.
There is no CPU path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import _lib, train_ops as T
from ._lib import HcirError, check
from .vit_engine import VitSpec


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _pad64(m: int) -> int:
    return (m + 63) // 64 * 64


class _Saved:
    """Activations of one training forward (device buffers; rows padded to a multiple of 64 with zeros)."""
    __slots__ = ("b", "t", "m", "x_in", "qkv", "att", "lse", "x_mid", "u", "ln1", "ln2", "h", "x_out", "patches", "last")


class VitTrainer:
    """fp16 operand copies of a ViT's weights (as stored and transposed) + the training forward / backward."""

    def __init__(self, spec: VitSpec, device: torch.device, keep_recomputable: bool = True,
                 cls_only_last: bool = True):
        self.keep_recomputable = bool(keep_recomputable)
        # forward() returns the class token only, so the last block's other rows are dead work (see forward)
        self.cls_only_last = bool(cls_only_last)
        if device.type != "cuda":
            raise HcirError(f"VitTrainer needs a HIP device, got {device} (no CPU fallback)")
        if spec.dim % spec.heads or spec.dim // spec.heads != 64:
            raise HcirError("hcir attention kernels support head_dim 64")
        if any(l.ls1 is not None or l.ls2 is not None for l in spec.layers):
            raise NotImplementedError("LayerScale is not on the training path")
        self.L = _lib.lib()
        self.device = device
        self.spec = spec
        self.dim, self.heads, self.eps, self.pos_mult = spec.dim, spec.heads, float(spec.eps), float(spec.pos_mult)
        self.patch = int(spec.patch)
        self.mlp = spec.layers[0].fc1_w.shape[0]
        if self.dim % 256 or self.mlp % 256 or (3 * spec.conv_w.shape[2] * spec.conv_w.shape[3]) % 256:
            raise HcirError("training path needs dim, mlp and C*P*P to be multiples of 256 (hcir_gemm_f16_tn)")
        self._key = None
        self.refresh()

    # -- parameters in the order the autograd Function sees them ------------------------------------------------
    def params(self) -> List[torch.Tensor]:
        s = self.spec
        out = [s.conv_w, s.conv_b, s.cls, s.pos]
        for l in s.layers:
            out += [l.ln1_w, l.ln1_b, l.qkv_w, l.qkv_b, l.proj_w, l.proj_b, l.ln2_w, l.ln2_b, l.fc1_w, l.fc1_b,
                    l.fc2_w, l.fc2_b]
        out += [s.final_ln_w, s.final_ln_b]
        return out

    def refresh(self) -> None:
        """(Re)build the fp16 operand copies when a parameter changed (optimizer step, load_state_dict)."""
        ps = self.params()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key == self._key:
            return
        dev, s = self.device, self.spec
        f16 = lambda t: t.detach().to(device=dev, dtype=torch.float16).contiguous()
        f16t = lambda t: t.detach().to(device=dev, dtype=torch.float16).t().contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.conv_w = f16(s.conv_w.reshape(self.dim, -1))
        self.conv_b, self.cls, self.pos = f32(s.conv_b), f32(s.cls.reshape(-1)), f32(s.pos.reshape(-1, self.dim))
        self.lw = []
        for l in s.layers:
            if l.qkv_b is None:
                raise NotImplementedError("qkv without bias is not on the training path")
            self.lw.append(dict(
                ln1_w=f32(l.ln1_w), ln1_b=f32(l.ln1_b), ln2_w=f32(l.ln2_w), ln2_b=f32(l.ln2_b),
                qkv_w=f16(l.qkv_w), qkv_wt=f16t(l.qkv_w), qkv_b=f32(l.qkv_b),
                proj_w=f16(l.proj_w), proj_wt=f16t(l.proj_w), proj_b=f32(l.proj_b),
                fc1_w=f16(l.fc1_w), fc1_wt=f16t(l.fc1_w), fc1_b=f32(l.fc1_b),
                fc2_w=f16(l.fc2_w), fc2_wt=f16t(l.fc2_w), fc2_b=f32(l.fc2_b)))
        self.fln_w, self.fln_b = f32(s.final_ln_w), f32(s.final_ln_b)
        self._key = key

    # -- small helpers over the C ABI ---------------------------------------------------------------------------
    def _gemm(self, a, k, w, bias, m, n, epi, out, st, what, resid=None):
        check(self.L.hcir_gemm_f16_resid(a.data_ptr(), k, w.data_ptr(), k, _p(bias), None, m, n, k, epi, _p(resid),
                                         out.data_ptr(), n, st), what)

    def _ln(self, x, rows, d, ldx, g, b, y, st):
        check(self.L.hcir_layernorm_f16(x.data_ptr(), _lib.F16, rows, d, ldx, g.data_ptr(), b.data_ptr(), self.eps,
                                        y.data_ptr(), d, st), "hcir_layernorm_f16")

    # -- forward ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise HcirError(f"input is on {x.device}; the hcir ViT runs on a HIP device only")
        self.refresh()
        x = x.float().contiguous()
        b, c, hh, ww = x.shape
        ps, d, L, dev = self.patch, self.dim, self.L, self.device
        if hh % ps or ww % ps:
            raise HcirError("image sides must be multiples of the patch size")
        t = (hh // ps) * (ww // ps) + 1
        if t > 256:
            raise HcirError("hcir_attn_bwd supports up to 256 tokens")
        if t != self.pos.shape[0]:
            raise HcirError(f"image gives {t} tokens but pos_embedding has {self.pos.shape[0]}")
        m, mp = b * t, _pad64(b * t)
        st = torch.cuda.current_stream(dev).cuda_stream
        def z16(cols):
            """[mp, cols] fp16 whose pad rows (>= m) are zero: kernels only ever write rows < m."""
            buf = torch.empty((mp, cols), dtype=torch.float16, device=dev)
            if mp > m:
                buf[m:].zero_()
            return buf

        sv = _Saved()
        sv.b, sv.t, sv.m = b, t, m
        sv.x_in, sv.qkv, sv.att, sv.lse, sv.x_mid, sv.u, sv.ln1, sv.ln2, sv.h = ([] for _ in range(9))
        # im2col of the patches in (c, ky, kx) order: the weight gradient of conv_proj needs it (data movement only)
        gh, gw = hh // ps, ww // ps
        pm = _pad64(b * gh * gw)
        sv.patches = torch.zeros((pm, c * ps * ps), dtype=torch.float16, device=dev)
        sv.patches[: b * gh * gw] = x.reshape(b, c, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(b * gh * gw, -1)
        tok = z16(d)
        kpad = self.conv_w.shape[1]
        if kpad % 64:
            raise HcirError("C*P*P must be a multiple of 64")
        check(L.hcir_patch_embed(x.data_ptr(), b, c, hh, ww, ps, self.conv_w.data_ptr(), kpad, self.conv_b.data_ptr(),
                                 self.cls.data_ptr(), self.pos.data_ptr(), self.pos_mult, d, tok.data_ptr(), _lib.F16,
                                 st), "hcir_patch_embed")
        scale = (d // self.heads) ** -0.5
        cur = tok
        # The LayerNorm outputs and gelu(u) are KEPT as well (they are the A operands of the weight-gradient GEMMs):
        # 3.6 KB more per token and layer - 65 GB for config C3's three 1024-image forwards, of 288 - instead of two
        # LayerNorm and one GELU launch per layer in the backward (keep_recomputable=False restores the recompute).
        keep = self.keep_recomputable
        ln, h = (None, None) if keep else (z16(d), z16(self.mlp))
        sv.last = None
        for li, w in enumerate(self.lw):
            qkv = z16(3 * d)
            if keep:
                ln = z16(d)
            self._ln(cur, m, d, d, w["ln1_w"], w["ln1_b"], ln, st)
            if keep:
                sv.ln1.append(ln)
            self._gemm(ln, d, w["qkv_w"], w["qkv_b"], m, 3 * d, _lib.EPI_BIAS_F16, qkv, st, "hcir_gemm_f16(qkv)")
            if self.cls_only_last and li == len(self.lw) - 1:
                # Only the class token feeds the loss (HP/src/main_backbone.py:625-627): in the LAST block the other
                # queries, and proj / LN2 / MLP of their rows, are dead in the forward and in the backward.  K and V
                # of every token are still needed (the qkv GEMM above); everything behind runs on the b class-token
                # rows, compact buffers [pad64(b), .] (as the inference engine has done since round 1).
                bp = _pad64(b)

                def c16(cols):
                    buf = torch.empty((bp, cols), dtype=torch.float16, device=dev)
                    if bp > b:
                        buf[b:].zero_()
                    return buf
                att_c, xin_c, x_mid_c, ln2_c, u_c, h_c, x_out_c = (c16(d), c16(d), c16(d), c16(d), c16(self.mlp),
                                                                   c16(self.mlp), c16(d))
                lse_c = torch.empty((b, self.heads), dtype=torch.float32, device=dev)
                check(L.hcir_attn_cls_fwd_lse(qkv.data_ptr(), b, t, self.heads, d // self.heads, scale, att_c.data_ptr(),
                                              lse_c.data_ptr(), st), "hcir_attn_cls_fwd_lse")
                xin_c[:b] = cur[:m].view(b, t, d)[:, 0]
                self._gemm(att_c, d, w["proj_w"], w["proj_b"], b, d, _lib.EPI_BIAS_RESID_F16, x_mid_c, st,
                           "hcir_gemm_f16(proj, cls rows)", resid=xin_c)
                self._ln(x_mid_c, b, d, d, w["ln2_w"], w["ln2_b"], ln2_c, st)
                if L.hcir_gemm_fused_supported(b, self.mlp, d):
                    check(L.hcir_gemm_f16_gelu_dual(ln2_c.data_ptr(), d, w["fc1_w"].data_ptr(), d, _p(w["fc1_b"]), b,
                                                    self.mlp, d, u_c.data_ptr(), h_c.data_ptr(), self.mlp, st),
                          "hcir_gemm_f16_gelu_dual(cls rows)")
                else:
                    self._gemm(ln2_c, d, w["fc1_w"], w["fc1_b"], b, self.mlp, _lib.EPI_BIAS_F16, u_c, st,
                               "hcir_gemm_f16(fc1, cls rows)")
                    check(L.hcir_gelu_fwd_f16(u_c.data_ptr(), b * self.mlp, h_c.data_ptr(), st), "hcir_gelu_fwd_f16")
                self._gemm(h_c, self.mlp, w["fc2_w"], w["fc2_b"], b, d, _lib.EPI_BIAS_RESID_F16, x_out_c, st,
                           "hcir_gemm_f16(fc2, cls rows)", resid=x_mid_c)
                sv.last = dict(att=att_c, lse=lse_c, x_mid=x_mid_c, ln2=ln2_c, u=u_c, h=h_c, x_out=x_out_c)
                sv.x_in.append(cur)
                sv.qkv.append(qkv)
                sv.x_out = None
                cls = torch.empty((b, d), dtype=torch.float32, device=dev)
                check(L.hcir_cls_head(x_out_c.data_ptr(), _lib.F16, b, 1, d, self.fln_w.data_ptr(), self.fln_b.data_ptr(),
                                      self.eps, 0, cls.data_ptr(), None, st), "hcir_cls_head")
                return cls, sv
            att = z16(d)
            lse = torch.empty((b, self.heads, t), dtype=torch.float32, device=dev)
            T.attn_fwd_lse(qkv, b, t, self.heads, scale, att, lse)
            x_mid = z16(d)       # out of place: `cur` is this block's saved input
            self._gemm(att, d, w["proj_w"], w["proj_b"], m, d, _lib.EPI_BIAS_RESID_F16, x_mid, st, "hcir_gemm_f16(proj)",
                       resid=cur)
            u = z16(self.mlp)
            if keep:
                ln, h = z16(d), z16(self.mlp)
                sv.ln2.append(ln)
                sv.h.append(h)
            self._ln(x_mid, m, d, d, w["ln2_w"], w["ln2_b"], ln, st)
            if L.hcir_gemm_fused_supported(m, self.mlp, d):
                # pre-activation (kept for the GELU backward) and activation from one pass over the accumulators
                check(L.hcir_gemm_f16_gelu_dual(ln.data_ptr(), d, w["fc1_w"].data_ptr(), d, _p(w["fc1_b"]), m, self.mlp,
                                                d, u.data_ptr(), h.data_ptr(), self.mlp, st), "hcir_gemm_f16_gelu_dual")
            else:
                self._gemm(ln, d, w["fc1_w"], w["fc1_b"], m, self.mlp, _lib.EPI_BIAS_F16, u, st, "hcir_gemm_f16(fc1)")
                check(L.hcir_gelu_fwd_f16(u.data_ptr(), m * self.mlp, h.data_ptr(), st), "hcir_gelu_fwd_f16")
            x_out = z16(d)
            self._gemm(h, self.mlp, w["fc2_w"], w["fc2_b"], m, d, _lib.EPI_BIAS_RESID_F16, x_out, st,
                       "hcir_gemm_f16(fc2)", resid=x_mid)
            for lst, v in ((sv.x_in, cur), (sv.qkv, qkv), (sv.att, att), (sv.lse, lse), (sv.x_mid, x_mid), (sv.u, u)):
                lst.append(v)
            cur = x_out
        sv.x_out = cur
        cls = torch.empty((b, d), dtype=torch.float32, device=dev)
        check(L.hcir_cls_head(cur.data_ptr(), _lib.F16, b, t, d, self.fln_w.data_ptr(), self.fln_b.data_ptr(), self.eps,
                              0, cls.data_ptr(), None, st), "hcir_cls_head")
        return cls, sv

    # -- backward -----------------------------------------------------------------------------------------------
    def backward(self, sv: _Saved, d_cls: torch.Tensor) -> List[torch.Tensor]:
        """Gradients (fp32, shapes of `params()`) of sum(cls * d_cls)."""
        b, t, m, d, dev, L = sv.b, sv.t, sv.m, self.dim, self.device, self.L
        mp = _pad64(m)
        st = torch.cuda.current_stream(dev).cuda_stream
        f32z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        # every gradient buffer is WRITTEN (accumulate=False) by the kernel that produces it: no zero fill (12 fills
        # per block); the [mp, *] fp16 work buffers only need their pad rows m.. zero (operands of the TN GEMM)
        f32e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)

        def pad16(cols):
            buf = torch.empty((mp, cols), dtype=torch.float16, device=dev)
            buf[m:].zero_()
            return buf
        grads = {}
        dres = f32z(mp, d)                                    # dL / d(residual stream), fp32
        dy16 = pad16(d)
        g_flw, g_flb = f32e(d), f32e(d)
        dcls16 = d_cls.detach().to(device=dev, dtype=torch.float16).contiguous()
        lastc = sv.last          # class-token-only last block: its residual gradient lives in a compact [pad64(b), d] buffer
        if lastc is None:
            T.layernorm_bwd(sv.x_out, dcls16, self.fln_w, self.eps, None, dres, g_flw, g_flb, accumulate=False, rows=b,
                            ldx=t * d, ldr=t * d)
        else:
            bp = _pad64(b)
            dres_c = f32z(bp, d)
            T.layernorm_bwd(lastc["x_out"], dcls16, self.fln_w, self.eps, None, dres_c, g_flw, g_flb, accumulate=False,
                            rows=b, ldx=d, ldr=d)
        big16 = pad16(self.mlp)                                                   # d_h / d_u
        # gelu(u) / LayerNorm output: recomputed per block only when the forward did not keep them
        hbuf = None if sv.h else pad16(self.mlp)
        lnbuf = None if sv.ln1 else pad16(d)
        dqkv = pad16(3 * d)
        datt = pad16(d)
        dln = pad16(d)
        layer_grads = []
        scale = (d // self.heads) ** -0.5
        nelem = m * d
        # dy16 = fp16(dres) and its column sums (the bias gradient of the Linear in front) come out of the LayerNorm
        # backward that produced dres (hcir_layernorm_bwd_fused) - except for the last block, whose dres is the
        # gradient of the final LayerNorm (class-token rows only)
        fc2_b_next = None
        for li in range(len(self.lw) - 1, -1, -1):
            w = self.lw[li]
            g = {k: f32e(*shape) for k, shape in (
                ("ln1_w", (d,)), ("ln1_b", (d,)), ("qkv_w", (3 * d, d)), ("qkv_b", (3 * d,)), ("proj_w", (d, d)),
                ("proj_b", (d,)), ("ln2_w", (d,)), ("ln2_b", (d,)), ("fc1_w", (self.mlp, d)), ("fc1_b", (self.mlp,)),
                ("fc2_w", (d, self.mlp)), ("fc2_b", (d,)))}
            if lastc is not None and li == len(self.lw) - 1:
                # ---- last block on the b class-token rows (compact buffers); K / V gradients for every token
                z16c = lambda cols: torch.zeros((bp, cols), dtype=torch.float16, device=dev)
                dy16_c, big16_c, dln_c, datt_c = z16c(d), z16c(self.mlp), z16c(d), z16c(d)
                check(L.hcir_add_f32_f16(dres_c.data_ptr(), None, b * d, dy16_c.data_ptr(), st), "hcir_add_f32_f16")
                T.colsum(dy16_c, g["fc2_b"], accumulate=False, rows=b)
                self._gemm(dy16_c, d, w["fc2_wt"], None, b, self.mlp, _lib.EPI_BIAS_F16, big16_c, st, "dgrad(fc2, cls)")
                T.gemm_tn(dy16_c, lastc["h"], g["fc2_w"], accumulate=False)
                T.gelu_bwd_colsum(lastc["u"], big16_c, big16_c, g["fc1_b"], rows=b)
                self._gemm(big16_c, self.mlp, w["fc1_wt"], None, b, d, _lib.EPI_BIAS_F16, dln_c, st, "dgrad(fc1, cls)")
                T.gemm_tn(big16_c, lastc["ln2"], g["fc1_w"], accumulate=False)
                T.layernorm_bwd(lastc["x_mid"], dln_c, w["ln2_w"], self.eps, dres_c, dres_c, g["ln2_w"], g["ln2_b"],
                                accumulate=False, rows=b, dres16=dy16_c, dres_colsum=g["proj_b"])
                self._gemm(dy16_c, d, w["proj_wt"], None, b, d, _lib.EPI_BIAS_F16, datt_c, st, "dgrad(proj, cls)")
                T.gemm_tn(dy16_c, lastc["att"], g["proj_w"], accumulate=False)
                check(L.hcir_attn_cls_bwd(sv.qkv[li].data_ptr(), lastc["att"].data_ptr(), datt_c.data_ptr(),
                                          lastc["lse"].data_ptr(), b, t, self.heads, d // self.heads, scale,
                                          dqkv.data_ptr(), st), "hcir_attn_cls_bwd")
                # the block's residual gradient: the class-token rows carry dres_c, every other row nothing (yet)
                dres[:m].view(b, t, d)[:, 0] = dres_c[:b]
                self._gemm(dqkv, 3 * d, w["qkv_wt"], None, m, d, _lib.EPI_BIAS_F16, dln, st, "dgrad(qkv)")
                if sv.ln1:
                    lncur = sv.ln1[li]
                else:
                    lncur = lnbuf
                    self._ln(sv.x_in[li], m, d, d, w["ln1_w"], w["ln1_b"], lnbuf, st)
                T.gemm_tn(dqkv, lncur, g["qkv_w"], accumulate=False)
                T.colsum(dqkv, g["qkv_b"], accumulate=False, rows=m)
                fc2_b_next = f32e(d) if li > 0 else None
                T.layernorm_bwd(sv.x_in[li], dln, w["ln1_w"], self.eps, dres, dres, g["ln1_w"], g["ln1_b"],
                                accumulate=False, rows=m, dres16=dy16 if li > 0 else None, dres_colsum=fc2_b_next)
                layer_grads.append(g)
                continue
            # ---- MLP: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
            if fc2_b_next is None:
                check(L.hcir_add_f32_f16(dres.data_ptr(), None, nelem, dy16.data_ptr(), st), "hcir_add_f32_f16")
                T.colsum(dy16, g["fc2_b"], accumulate=False, rows=m)
            else:
                g["fc2_b"] = fc2_b_next
            self._gemm(dy16, d, w["fc2_wt"], None, m, self.mlp, _lib.EPI_BIAS_F16, big16, st, "dgrad(fc2)")
            if sv.h:
                hcur = sv.h[li]
            else:
                hcur = hbuf
                check(L.hcir_gelu_fwd_f16(sv.u[li].data_ptr(), m * self.mlp, hbuf.data_ptr(), st), "hcir_gelu_fwd_f16")
            T.gemm_tn(dy16, hcur, g["fc2_w"], accumulate=False)
            T.gelu_bwd_colsum(sv.u[li], big16, big16, g["fc1_b"], rows=m)     # d_u in place, + the fc1 bias gradient
            self._gemm(big16, self.mlp, w["fc1_wt"], None, m, d, _lib.EPI_BIAS_F16, dln, st, "dgrad(fc1)")
            if sv.ln2:
                lncur = sv.ln2[li]
            else:
                lncur = lnbuf
                self._ln(sv.x_mid[li], m, d, d, w["ln2_w"], w["ln2_b"], lnbuf, st)
            T.gemm_tn(big16, lncur, g["fc1_w"], accumulate=False)
            T.layernorm_bwd(sv.x_mid[li], dln, w["ln2_w"], self.eps, dres, dres, g["ln2_w"], g["ln2_b"],
                            accumulate=False, rows=m, dres16=dy16, dres_colsum=g["proj_b"])
            # ---- attention: x_mid = x_in + proj(attn(qkv(LN1(x_in))))
            self._gemm(dy16, d, w["proj_wt"], None, m, d, _lib.EPI_BIAS_F16, datt, st, "dgrad(proj)")
            T.gemm_tn(dy16, sv.att[li], g["proj_w"], accumulate=False)
            T.attn_bwd(sv.qkv[li], sv.att[li], datt, sv.lse[li], b, t, self.heads, scale, dqkv)
            self._gemm(dqkv, 3 * d, w["qkv_wt"], None, m, d, _lib.EPI_BIAS_F16, dln, st, "dgrad(qkv)")
            if sv.ln1:
                lncur = sv.ln1[li]
            else:
                lncur = lnbuf
                self._ln(sv.x_in[li], m, d, d, w["ln1_w"], w["ln1_b"], lnbuf, st)
            T.gemm_tn(dqkv, lncur, g["qkv_w"], accumulate=False)
            T.colsum(dqkv, g["qkv_b"], accumulate=False, rows=m)
            fc2_b_next = f32e(d) if li > 0 else None
            T.layernorm_bwd(sv.x_in[li], dln, w["ln1_w"], self.eps, dres, dres, g["ln1_w"], g["ln1_b"],
                            accumulate=False, rows=m, dres16=dy16 if li > 0 else None, dres_colsum=fc2_b_next)
            layer_grads.append(g)
        layer_grads.reverse()
        # ---- patch embedding: tok[b][0] = cls + pos_mult pos[0]; tok[b][1+p] = W patch + bias + pos_mult pos[1+p]
        dtok = dres[:m].view(b, t, d)
        g_cls = dtok[:, 0].sum(0)
        g_pos = dtok.sum(0) * self.pos_mult
        npat = b * (t - 1)
        dpatch = torch.zeros((_pad64(npat), d), dtype=torch.float16, device=dev)
        dpatch[:npat] = dtok[:, 1:].reshape(npat, d)
        g_cw = f32e(d, sv.patches.shape[1])
        g_cb = f32e(d)
        T.gemm_tn(dpatch, sv.patches, g_cw, accumulate=False)
        T.colsum(dpatch, g_cb, accumulate=False, rows=npat)
        s = self.spec
        out = [g_cw.view(s.conv_w.shape), g_cb, g_cls.view(s.cls.shape), g_pos.view(s.pos.shape)]
        for g in layer_grads:
            out += [g["ln1_w"], g["ln1_b"], g["qkv_w"], g["qkv_b"], g["proj_w"], g["proj_b"], g["ln2_w"], g["ln2_b"],
                    g["fc1_w"], g["fc1_b"], g["fc2_w"], g["fc2_b"]]
        out += [g_flw, g_flb]
        return out


class _VitClsFn(torch.autograd.Function):
    """cls = LN_final(ViT(x))[:, 0] with gradients for every backbone parameter (none for the images)."""

    @staticmethod
    def forward(ctx, trainer: VitTrainer, x: torch.Tensor, *params):
        cls, saved = trainer.forward(x)
        ctx.trainer, ctx.saved = trainer, saved
        ctx.mask = [p.requires_grad for p in params]
        return cls

    @staticmethod
    def backward(ctx, d_cls):
        # The backward's GEMM operands are fp16.  An incoming gradient far from 1 (no GradScaler: d_cls ~ 1e-5 at
        # batch 1024; or a loss scale of 2^24) would push dS = P (dP - D) and the dgrad operands into fp16's subnormal
        # / overflow range, so the gradient is renormalised here by a POWER OF TWO (exact in fp32 and fp16) to
        # max|d_cls| in [0.5, 1), and the fp32 results are scaled back.  Device-side only (no host sync); a
        # non-finite or all-zero d_cls passes through unscaled, so an overflowed GradScaler step still shows infs.
        amax = d_cls.detach().abs().max().float()
        ok = torch.isfinite(amax) & (amax > 0)
        k = torch.where(ok, -torch.floor(torch.log2(torch.where(ok, amax, torch.ones_like(amax)))) - 1.0,
                        torch.zeros_like(amax)).clamp_(-100.0, 100.0)
        up, down = torch.exp2(k), torch.exp2(-k)
        grads = ctx.trainer.backward(ctx.saved, d_cls * up)
        ctx.saved = None
        torch._foreach_mul_(grads, down)
        return (None, None) + tuple(g if need else None for g, need in zip(grads, ctx.mask))


def vit_cls_with_grad(trainer: VitTrainer, x: torch.Tensor) -> torch.Tensor:
    return _VitClsFn.apply(trainer, x, *trainer.params())
