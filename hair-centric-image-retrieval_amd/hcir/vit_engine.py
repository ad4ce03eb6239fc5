"""hcir.vit_engine — the ViT forward on MI355X through libhcir (C ABI, HIP kernels).

A `VitSpec` is the architecture-neutral description both model families
(torchvision-layout ViTWrapper, timm-layout models_vit.VisionTransformer) reduce to.
`VitEngine.forward_tokens` runs
    patch_embed (+cls, +pos)  ->  depth x [ LN -> qkv GEMM -> fused attention ->
    proj GEMM (+residual) -> LN -> fc1 GEMM (+GELU) -> fc2 GEMM (+residual) ]
with fp16 MFMA operands, fp32 accumulation and an fp32 or fp16 residual stream, entirely in
device buffers the engine owns (re-used across calls).  With the fp16 stream and >= 1024 token rows
the LayerNorms are folded into the neighbouring GEMMs (hcir_gemm_f16_fused).  There is no CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib
from ._lib import HcirError, check


# Residual-stream storage of engines built without an explicit choice.  fp32 is the conservative
# default; set to torch.float16 (or HCIR_RESID=f16) to halve LayerNorm / residual-epilogue traffic.
import os as _os

DEFAULT_RESID_DTYPE = torch.float16 if _os.environ.get("HCIR_RESID", "f32") == "f16" else torch.float32
# LayerNorm folded into the neighbouring GEMMs (fp16 residual stream, persistent-kernel shapes only);
# HCIR_LN_FUSE=0 keeps the separate hcir_layernorm_f16 launches
LN_FUSE = _os.environ.get("HCIR_LN_FUSE", "1") != "0"


@dataclass
class VitLayer:
    ln1_w: torch.Tensor
    ln1_b: torch.Tensor
    qkv_w: torch.Tensor   # [3D, D]
    qkv_b: Optional[torch.Tensor]
    proj_w: torch.Tensor  # [D, D]
    proj_b: torch.Tensor
    ln2_w: torch.Tensor
    ln2_b: torch.Tensor
    fc1_w: torch.Tensor   # [mlp, D]
    fc1_b: torch.Tensor
    fc2_w: torch.Tensor   # [D, mlp]
    fc2_b: torch.Tensor
    ls1: Optional[torch.Tensor] = None  # LayerScale gamma
    ls2: Optional[torch.Tensor] = None


@dataclass
class VitSpec:
    patch: int
    dim: int
    heads: int
    eps: float
    pos_mult: float           # 2.0 for the reference's ViTWrapper (double positional add)
    conv_w: torch.Tensor      # [D, 3, P, P]
    conv_b: torch.Tensor
    cls: torch.Tensor         # [1, 1, D]
    pos: torch.Tensor         # [1, T, D]
    layers: List[VitLayer] = field(default_factory=list)
    final_ln_w: Optional[torch.Tensor] = None
    final_ln_b: Optional[torch.Tensor] = None


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _f16(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float16).contiguous()


class _DevLayer:
    def __init__(self, l: VitLayer, dev):
        self.ln1_w, self.ln1_b = _f32(l.ln1_w, dev), _f32(l.ln1_b, dev)
        self.ln2_w, self.ln2_b = _f32(l.ln2_w, dev), _f32(l.ln2_b, dev)
        self.qkv_w, self.proj_w = _f16(l.qkv_w, dev), _f16(l.proj_w, dev)
        self.fc1_w, self.fc2_w = _f16(l.fc1_w, dev), _f16(l.fc2_w, dev)
        self.qkv_b = None if l.qkv_b is None else _f32(l.qkv_b, dev)
        self.proj_b, self.fc1_b, self.fc2_b = _f32(l.proj_b, dev), _f32(l.fc1_b, dev), _f32(l.fc2_b, dev)
        self.ls1 = None if l.ls1 is None else _f32(l.ls1, dev)
        self.ls2 = None if l.ls2 is None else _f32(l.ls2, dev)
        # LayerNorm fold (hcir_gemm_f16_fused): W' = fp16(gamma o W), c1 = row sums of the ROUNDED W',
        # bias' = W . beta + b.  LN(x) W^T + b == rstd (x W'^T - mean c1) + bias'
        self.qkv_wg, self.qkv_c1, self.qkv_c2 = self._fold(l.qkv_w, l.ln1_w, l.ln1_b, l.qkv_b, dev)
        self.fc1_wg, self.fc1_c1, self.fc1_c2 = self._fold(l.fc1_w, l.ln2_w, l.ln2_b, l.fc1_b, dev)

    @staticmethod
    def _fold(w, gamma, beta, bias, dev):
        w64 = w.detach().to(device=dev, dtype=torch.float64)
        g64 = gamma.detach().to(device=dev, dtype=torch.float64)
        b64 = beta.detach().to(device=dev, dtype=torch.float64)
        wg = (w64 * g64[None, :]).to(torch.float16).contiguous()
        c1 = wg.to(torch.float64).sum(1).to(torch.float32).contiguous()
        c2 = w64 @ b64
        if bias is not None:
            c2 = c2 + bias.detach().to(device=dev, dtype=torch.float64)
        return wg, c1, c2.to(torch.float32).contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class VitEngine:
    """Device-resident fp16 weights + work buffers for one ViT; forward via the C ABI."""

    def __init__(self, spec: VitSpec, device: torch.device, resid_dtype: torch.dtype = None):
        # storage type of the residual stream: fp32 (conservative) or fp16 (half the LayerNorm /
        # residual-epilogue HBM traffic; rounding 2^-11 per update ~ 1e-6 in embedding cosine)
        if resid_dtype is None:
            resid_dtype = DEFAULT_RESID_DTYPE
        if resid_dtype not in (torch.float32, torch.float16):
            raise HcirError("resid_dtype must be torch.float32 or torch.float16")
        self.resid_dtype = resid_dtype
        self._rt = _lib.F32 if resid_dtype == torch.float32 else _lib.F16
        self._resid_epi = _lib.EPI_BIAS_RESID_F32 if resid_dtype == torch.float32 else _lib.EPI_BIAS_RESID_F16
        if device.type != "cuda":
            raise HcirError(f"VitEngine needs a HIP device, got {device} (no CPU fallback)")
        if spec.dim % spec.heads or spec.dim // spec.heads not in (32, 48, 64, 80, 96, 128):
            raise HcirError("hcir_attn_fwd supports head_dim 32, 48, 64 (tuned), 80, 96, 128")
        self.L = _lib.lib()
        self.device = device
        self.dim, self.heads, self.eps, self.pos_mult = spec.dim, spec.heads, float(spec.eps), float(spec.pos_mult)
        self.patch = int(spec.patch)
        cw = _f16(spec.conv_w.reshape(spec.dim, -1), device)
        kpad = (cw.shape[1] + 63) // 64 * 64           # rows zero-padded to the 64-wide k chunk
        self.conv_w = torch.zeros((spec.dim, kpad), dtype=torch.float16, device=device)
        self.conv_w[:, : cw.shape[1]] = cw
        self.conv_b = _f32(spec.conv_b, device)
        self.cls = _f32(spec.cls.reshape(-1), device)
        self.pos = _f32(spec.pos.reshape(-1, spec.dim), device)
        self.layers = [_DevLayer(l, device) for l in spec.layers]
        self.mlp = spec.layers[0].fc1_w.shape[0] if spec.layers else 4 * spec.dim
        self.fln_w = None if spec.final_ln_w is None else _f32(spec.final_ln_w, device)
        self.fln_b = None if spec.final_ln_b is None else _f32(spec.final_ln_b, device)
        self._bufs = {}
        self.prof = None  # optional hcir.profiling.EventProfiler (per-op HIP events)

    def _mark(self, name: str) -> None:
        if self.prof is not None:
            self.prof.mark(name)

    # -- buffers ---------------------------------------------------------------
    def _buffers(self, b: int, t: int, slot: int = 0):
        """Work buffers of one forward in flight.  `slot`: which of several forwards in flight at once (one per HIP
        stream, hcir.pipeline.StreamPipeline); buffers of another shape are dropped."""
        key = (b, t, slot)
        bufs = self._bufs.get(key)
        if bufs is None:
            m, d, dev = b * t, self.dim, self.device
            bufs = dict(
                tok=torch.empty((b, t, d), dtype=self.resid_dtype, device=dev),
                ln=torch.empty((m, d), dtype=torch.float16, device=dev),
                qkv=torch.empty((m, 3 * d), dtype=torch.float16, device=dev),
                att=torch.empty((m, d), dtype=torch.float16, device=dev),
                hid=torch.empty((m, self.mlp), dtype=torch.float16, device=dev),
                # compact buffers of the CLS-only last block (B rows)
                att_c=torch.empty((b, d), dtype=torch.float16, device=dev),
                ln_c=torch.empty((b, d), dtype=torch.float16, device=dev),
                hid_c=torch.empty((b, self.mlp), dtype=torch.float16, device=dev),
                # LayerNorm fold: partial (sum, sumsq) slices of the residual rows and their (mean, rstd)
                stats_part=torch.empty((max(d // 64, 1), m, 2), dtype=torch.float32, device=dev),
                stats=torch.empty((m, 2), dtype=torch.float32, device=dev),
            )
            self._bufs = {k: v for k, v in self._bufs.items() if k[:2] == (b, t)}  # keep one shape resident
            self._bufs[key] = bufs
        return bufs

    # -- forward ---------------------------------------------------------------
    def forward_tokens(self, x: torch.Tensor, cls_only_last: bool = False, slot: int = 0) -> torch.Tensor:
        """x fp32 [B,3,H,W] on the HIP device -> token buffer [B,T,D] in resid_dtype (engine-owned).

        cls_only_last: the caller consumes only the class token (extract_features).  The last block
        then computes K and V for every token but attention queries, proj, LN2 and the MLP for the
        class-token rows only (B rows instead of B*T): same arithmetic for those rows, ~1/12 of the
        forward saved.  Patch-token rows of the returned buffer are then those BEFORE the last block."""
        if not x.is_cuda:
            raise HcirError(f"input is on {x.device}; the hcir ViT runs on a HIP device only")
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        b, c, hh, ww = x.shape
        ps = self.patch
        if hh % ps or ww % ps:
            raise HcirError("image sides must be multiples of the patch size")
        t = (hh // ps) * (ww // ps) + 1
        if t > 288:
            raise HcirError("hcir_attn_fwd supports up to 288 tokens")
        if t != self.pos.shape[0]:
            raise HcirError(f"image gives {t} tokens but pos_embedding has {self.pos.shape[0]}")
        L, d, m = self.L, self.dim, b * t
        st = torch.cuda.current_stream(x.device).cuda_stream
        w = self._buffers(b, t, slot)
        tok, ln, qkv, att, hid = w["tok"], w["ln"], w["qkv"], w["att"], w["hid"]
        check(L.hcir_patch_embed(x.data_ptr(), b, c, hh, ww, ps, self.conv_w.data_ptr(), self.conv_w.shape[1],
                                 self.conv_b.data_ptr(), self.cls.data_ptr(), self.pos.data_ptr(),
                                 self.pos_mult, d, tok.data_ptr(), self._rt, st), "hcir_patch_embed")
        self._mark("patch_embed")
        scale = (d // self.heads) ** -0.5
        last = len(self.layers) - 1
        # LayerNorm fold: fp16 residual stream and every GEMM of the block on the persistent kernel
        fuse = (LN_FUSE and self.resid_dtype == torch.float16
                and L.hcir_gemm_fused_supported(m, 3 * d, d) and L.hcir_gemm_fused_supported(m, d, d)
                and L.hcir_gemm_fused_supported(m, self.mlp, d) and L.hcir_gemm_fused_supported(m, d, self.mlp))
        sp, stats = w["stats_part"], w["stats"]
        nsl = d // 64
        have_stats = False  # `stats` holds (mean, rstd) of the current residual rows

        def resid_gemm(a, k, wt, bias, ls, want_stats, what):
            if want_stats:
                check(L.hcir_gemm_f16_fused(a.data_ptr(), k, wt.data_ptr(), k, bias.data_ptr(), _p(ls), m, d, k,
                                            _lib.EPI_BIAS_RESID_F16, tok.data_ptr(), d, None, None, sp.data_ptr(),
                                            st), what)
                check(L.hcir_ln_stats_finalize(sp.data_ptr(), nsl, m, d, self.eps, stats.data_ptr(), st),
                      "hcir_ln_stats_finalize")
            else:
                check(L.hcir_gemm_f16(a.data_ptr(), k, wt.data_ptr(), k, bias.data_ptr(), _p(ls), m, d, k,
                                      self._resid_epi, tok.data_ptr(), d, st), what)

        for li, l in enumerate(self.layers):
            cls_only = cls_only_last and li == last
            if fuse and have_stats:
                check(L.hcir_gemm_f16_fused(tok.data_ptr(), d, l.qkv_wg.data_ptr(), d, l.qkv_c2.data_ptr(), None,
                                            m, 3 * d, d, _lib.EPI_BIAS_F16, qkv.data_ptr(), 3 * d,
                                            stats.data_ptr(), l.qkv_c1.data_ptr(), None, st),
                      "hcir_gemm_f16_fused(ln1+qkv)")
            else:
                check(L.hcir_layernorm_f16(tok.data_ptr(), self._rt, m, d, d, l.ln1_w.data_ptr(),
                                           l.ln1_b.data_ptr(), self.eps, ln.data_ptr(), d, st), "hcir_layernorm_f16")
                self._mark("layernorm")
                check(L.hcir_gemm_f16(ln.data_ptr(), d, l.qkv_w.data_ptr(), d, _p(l.qkv_b), None, m, 3 * d, d,
                                      _lib.EPI_BIAS_F16, qkv.data_ptr(), 3 * d, st), "hcir_gemm_f16(qkv)")
            have_stats = False
            self._mark("gemm_qkv")
            if not cls_only:
                check(L.hcir_attn_fwd(qkv.data_ptr(), b, t, self.heads, d // self.heads, scale, t,
                                      att.data_ptr(), st), "hcir_attn_fwd")
                self._mark("attn")
                resid_gemm(att, d, l.proj_w, l.proj_b, l.ls1, fuse, "hcir_gemm_f16(proj)")
                self._mark("gemm_proj")
                if fuse:
                    check(L.hcir_gemm_f16_fused(tok.data_ptr(), d, l.fc1_wg.data_ptr(), d, l.fc1_c2.data_ptr(), None,
                                                m, self.mlp, d, _lib.EPI_BIAS_GELU_F16, hid.data_ptr(), self.mlp,
                                                stats.data_ptr(), l.fc1_c1.data_ptr(), None, st),
                          "hcir_gemm_f16_fused(ln2+fc1)")
                else:
                    check(L.hcir_layernorm_f16(tok.data_ptr(), self._rt, m, d, d, l.ln2_w.data_ptr(),
                                               l.ln2_b.data_ptr(), self.eps, ln.data_ptr(), d, st),
                          "hcir_layernorm_f16")
                    self._mark("layernorm")
                    check(L.hcir_gemm_f16(ln.data_ptr(), d, l.fc1_w.data_ptr(), d, l.fc1_b.data_ptr(), None, m,
                                          self.mlp, d, _lib.EPI_BIAS_GELU_F16, hid.data_ptr(), self.mlp, st),
                          "hcir_gemm_f16(fc1)")
                self._mark("gemm_fc1")
                # the row statistics after fc2 feed the NEXT block's ln_1 (none after the last block)
                have_stats = fuse and li < last
                resid_gemm(hid, self.mlp, l.fc2_w, l.fc2_b, l.ls2, have_stats, "hcir_gemm_f16(fc2)")
                self._mark("gemm_fc2")
            else:
                # class-token rows only: row b of the compact buffers <-> tok[b][0] (row stride t*d)
                att_c, ln_c, hid_c = w["att_c"], w["ln_c"], w["hid_c"]
                check(L.hcir_attn_fwd(qkv.data_ptr(), b, t, self.heads, d // self.heads, scale, 1,
                                      att_c.data_ptr(), st), "hcir_attn_fwd(cls)")
                self._mark("attn")
                check(L.hcir_gemm_f16(att_c.data_ptr(), d, l.proj_w.data_ptr(), d, l.proj_b.data_ptr(), _p(l.ls1),
                                      b, d, d, self._resid_epi, tok.data_ptr(), t * d, st), "hcir_gemm_f16(proj,cls)")
                self._mark("gemm_proj")
                check(L.hcir_layernorm_f16(tok.data_ptr(), self._rt, b, d, t * d, l.ln2_w.data_ptr(),
                                           l.ln2_b.data_ptr(), self.eps, ln_c.data_ptr(), d, st),
                      "hcir_layernorm_f16(cls)")
                self._mark("layernorm")
                check(L.hcir_gemm_f16(ln_c.data_ptr(), d, l.fc1_w.data_ptr(), d, l.fc1_b.data_ptr(), None, b,
                                      self.mlp, d, _lib.EPI_BIAS_GELU_F16, hid_c.data_ptr(), self.mlp, st),
                      "hcir_gemm_f16(fc1,cls)")
                self._mark("gemm_fc1")
                check(L.hcir_gemm_f16(hid_c.data_ptr(), self.mlp, l.fc2_w.data_ptr(), self.mlp, l.fc2_b.data_ptr(),
                                      _p(l.ls2), b, d, self.mlp, self._resid_epi, tok.data_ptr(), t * d, st),
                      "hcir_gemm_f16(fc2,cls)")
                self._mark("gemm_fc2")
        return tok

    def cls_embedding(self, tok: torch.Tensor, final_norm: bool, l2_normalize: bool,
                      want_f16: bool = False):
        """CLS row of the token buffer, through the final LayerNorm (if the model has one)
        and optionally F.normalize; fp32 [B,D] (and an fp16 copy for the similarity scan)."""
        b, t, d = tok.shape
        st = torch.cuda.current_stream(tok.device).cuda_stream
        e32 = torch.empty((b, d), dtype=torch.float32, device=tok.device)
        e16 = torch.empty((b, d), dtype=torch.float16, device=tok.device) if want_f16 else None
        g = self.fln_w if final_norm else None
        bb = self.fln_b if final_norm else None
        if final_norm and g is None:
            raise HcirError("model has no final LayerNorm")
        check(self.L.hcir_cls_head(tok.data_ptr(), self._rt, b, t, d, _p(g), _p(bb), self.eps, int(l2_normalize),
                                   e32.data_ptr(), _p(e16), st), "hcir_cls_head")
        return (e32, e16) if want_f16 else e32

    def patch_mean(self, tok: torch.Tensor, final_norm: bool) -> torch.Tensor:
        b, t, d = tok.shape
        st = torch.cuda.current_stream(tok.device).cuda_stream
        out = torch.empty((b, d), dtype=torch.float32, device=tok.device)
        g = self.fln_w if final_norm else None
        bb = self.fln_b if final_norm else None
        check(self.L.hcir_patch_mean(tok.data_ptr(), self._rt, b, t, d, _p(g), _p(bb), self.eps, out.data_ptr(), st),
              "hcir_patch_mean")
        return out


class EngineCache:
    """Builds the VitEngine lazily and rebuilds it when parameters or device change.

    Every call fingerprints EVERY parameter (data_ptr, _version): load_state_dict, optimizer steps, .to() and
    in-place ops all bump one of the two (~50 us of host time per forward for ~150 tensors).  What it cannot see
    is a write through `p.data` / a raw pointer, which bumps neither: call `invalidate()` after such an edit."""

    def __init__(self):
        self._engine: Optional[VitEngine] = None
        self._key = None

    def invalidate(self) -> None:
        self._engine, self._key = None, None

    def get(self, params, make_spec, device: torch.device) -> VitEngine:
        key = (str(device), DEFAULT_RESID_DTYPE) + tuple((p.data_ptr(), p._version) for p in params)
        if self._engine is None or key != self._key:
            self._engine = VitEngine(make_spec(), device)
            self._key = key
        return self._engine
