"""hcir.main_backbone — HSimCLR model classes with the reference's constructor
signatures (HP/src/main_backbone.py:528-637), backed by the MI355X hot path.

  ViTWrapper(weights=None)             HP/src/main_backbone.py:528-563 (second definition,
                                       the one that wins at call time)
  SHAM2(model="resnet18")              HP/src/main_backbone.py:565-637
  SimCLRProjectionHead(in, hid, out)   lightly.models.modules (keys layers.{0,1,3,4})

ViT compute goes through hcir.vit_engine -> libhcir.so (hand-written HIP);
ResNet trunks are PyTorch-ROCm modules.  The ViT path has no CPU
fallback.  With autograd enabled on parameters that require grad, `forward_cls` runs the
training forward of hcir.vit_train (activations kept, backward through hcir_gemm_f16_tn /
hcir_attn_bwd / hcir_layernorm_bwd ...): `SHAM2.forward` in train mode is differentiable
(HP/src/pretrain_engine.py:682-745).  `ViTWrapper.forward` (class token AND pooled patches) stays
inference-only.
"""
from __future__ import annotations

import copy

import torch
from torch import nn

from . import _lib
from ._lib import HcirError, check
from . import _tv_resnet, _tv_vit
from .vit_engine import EngineCache, VitLayer, VitSpec
from .vit_train import VitTrainer, vit_cls_with_grad

# The reference adds the positional embedding in ViTWrapper.forward AND again inside
# torchvision's Encoder.forward (same Parameter): embeddings are LN(blocks(x + 2*pos)).
# Reproduced on purpose so reference checkpoints give identical embeddings
# (HP/src/main_backbone.py:537,551,554).
POS_EMBED_MULT = 2.0


def deactivate_requires_grad(module: nn.Module) -> None:
    """lightly.models.utils.deactivate_requires_grad."""
    for p in module.parameters():
        p.requires_grad = False


class SimCLRProjectionHead(nn.Module):
    """lightly SimCLRProjectionHead(input_dim, hidden_dim, output_dim), num_layers=2:
    Linear(no bias) -> BatchNorm1d -> ReLU -> Linear(no bias) -> BatchNorm1d
    (state-dict keys layers.0.weight, layers.1.*, layers.3.weight, layers.4.*)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, output_dim: int = 128):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Linear(input_dim, hidden_dim, bias=False), nn.BatchNorm1d(hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, output_dim, bias=False), nn.BatchNorm1d(output_dim))

    def _hip_trainable(self) -> bool:
        lin0, _, _, lin1, _ = self.layers
        return all(f % 256 == 0 for f in (lin0.in_features, lin0.out_features, lin1.out_features))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and x.is_cuda and self._hip_trainable():
            # train mode on the HIP path: batch-statistics BatchNorm kernels + hcir GEMMs, forward and backward
            # (hcir.head_train); the ViT head (768, 768, 512) and the ResNet-50 head (2048, 2048, 1024) qualify
            from .head_train import head_train_forward
            return head_train_forward(self, x)
        return self.layers(x)   # CPU tensors / the ResNet-18 head (128 outputs): torch modules, as the trunk is

    # eval-mode forward on the HIP device: two GEMMs with the BatchNorm folded into the
    # epilogue (scale = w / sqrt(var + eps), shift = b - mean * scale)
    def forward_hip(self, x16: torch.Tensor) -> torch.Tensor:
        if self.training:
            return self.forward(x16.float())
        L = _lib.lib()
        lin0, bn0, _, lin1, bn1 = self.layers
        dev = x16.device
        st = torch.cuda.current_stream(dev).cuda_stream

        def fold(bn):
            s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            return s, (bn.bias - bn.running_mean * s).float().contiguous()

        s0, b0 = fold(bn0)
        s1, b1 = fold(bn1)
        w0 = lin0.weight.detach().half().contiguous()
        w1 = lin1.weight.detach().half().contiguous()
        m, k0 = x16.shape
        hid = torch.empty((m, w0.shape[0]), dtype=torch.float16, device=dev)
        out = torch.empty((m, w1.shape[0]), dtype=torch.float32, device=dev)
        check(L.hcir_gemm_f16(x16.data_ptr(), k0, w0.data_ptr(), k0, b0.data_ptr(), s0.data_ptr(), m,
                              w0.shape[0], k0, _lib.EPI_AFFINE_RELU_F16, hid.data_ptr(), w0.shape[0], st),
              "hcir_gemm_f16(head0)")
        check(L.hcir_gemm_f16(hid.data_ptr(), w0.shape[0], w1.data_ptr(), w0.shape[0], b1.data_ptr(),
                              s1.data_ptr(), m, w1.shape[0], w0.shape[0], _lib.EPI_AFFINE_F32,
                              out.data_ptr(), w1.shape[0], st), "hcir_gemm_f16(head1)")
        return out


class ViTWrapper(nn.Module):
    """torchvision vit_b_16 without its head, as wrapped by the reference.

    forward(x) -> (cls_token [B,768], pooled_patches [B,768]) where both come from
    ln(encoder_blocks(conv_proj(x) ++ cls + 2 * pos_embedding)).
    """

    def __init__(self, weights=None):
        super().__init__()
        vit = _tv_vit.vit_b_16(weights=weights)
        self.conv_proj = vit.conv_proj
        self.encoder = vit.encoder
        self.cls_token = vit.class_token
        self.pos_embedding = vit.encoder.pos_embedding
        self._num_heads = vit.num_heads
        self._cache = EngineCache()

    # -- HIP engine ------------------------------------------------------------
    def _spec(self) -> VitSpec:
        layers = []
        for blk in self.encoder.layers:
            mha = blk.self_attention
            layers.append(VitLayer(
                ln1_w=blk.ln_1.weight, ln1_b=blk.ln_1.bias,
                qkv_w=mha.in_proj_weight, qkv_b=mha.in_proj_bias,
                proj_w=mha.out_proj.weight, proj_b=mha.out_proj.bias,
                ln2_w=blk.ln_2.weight, ln2_b=blk.ln_2.bias,
                fc1_w=blk.mlp[0].weight, fc1_b=blk.mlp[0].bias,
                fc2_w=blk.mlp[3].weight, fc2_b=blk.mlp[3].bias))
        return VitSpec(patch=16, dim=self.conv_proj.out_channels, heads=self._num_heads, eps=1e-6,
                       pos_mult=POS_EMBED_MULT, conv_w=self.conv_proj.weight, conv_b=self.conv_proj.bias,
                       cls=self.cls_token, pos=self.pos_embedding, layers=layers,
                       final_ln_w=self.encoder.ln.weight, final_ln_b=self.encoder.ln.bias)

    def engine(self, device: torch.device):
        return self._cache.get(list(self.parameters()), self._spec, device)

    def _needs_grad(self) -> bool:
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _check_no_grad(self):
        if self._needs_grad():
            raise NotImplementedError(
                "ViTWrapper.forward (class token + pooled patches) is inference-only on the HIP path: wrap the call "
                "in torch.no_grad(), or use forward_cls / SHAM2.forward, which are differentiable")

    def trainer(self, device: torch.device) -> VitTrainer:
        """Training-side twin of `engine`: fp16 operand copies (as stored and transposed) refreshed when a
        parameter's version changes (optimizer step), activations kept per forward."""
        tr = getattr(self, "_trainer", None)
        if tr is None or tr.device != device:
            tr = VitTrainer(self._spec(), device)
            self._trainer = tr
        return tr

    def forward(self, x: torch.Tensor):
        self._check_no_grad()
        eng = self.engine(x.device)
        tok = eng.forward_tokens(x)
        cls_token = eng.cls_embedding(tok, final_norm=True, l2_normalize=False)
        pooled_patches = eng.patch_mean(tok, final_norm=True)
        return cls_token, pooled_patches

    def forward_cls(self, x: torch.Tensor, l2_normalize: bool = False, want_f16: bool = False, slot: int = 0):
        """CLS embedding only (what SHAM2.extract_features consumes); optionally fused
        F.normalize and an fp16 copy for the similarity scan.  Differentiable when autograd is on and the
        parameters require grad (then fp32 [B, D], no fused normalise)."""
        if self._needs_grad():
            if l2_normalize or want_f16:
                raise NotImplementedError("the differentiable forward returns the plain fp32 class token")
            return vit_cls_with_grad(self.trainer(x.device), x)
        eng = self.engine(x.device)
        tok = eng.forward_tokens(x, cls_only_last=True, slot=slot)   # only the class token is consumed
        return eng.cls_embedding(tok, final_norm=True, l2_normalize=l2_normalize, want_f16=want_f16)


class SHAM2(nn.Module):
    """HSimCLR model (README 'HSimCLR' == class SHAM2, HP/src/main_backbone.py:565-637)."""

    def __init__(self, model="resnet18"):
        super().__init__()
        self.model = model
        print("Using backbone:", self.model)

        if model == "resnet18":
            backbone = _tv_resnet.resnet18(weights=None)
            self.backbone = nn.Sequential(*list(backbone.children())[:-1])
            proj_input_dim, output_dim = 512, 128
        elif model == "resnet50":
            backbone = _tv_resnet.resnet50(weights=None)
            self.backbone = nn.Sequential(*list(backbone.children())[:-1])
            proj_input_dim, output_dim = 2048, 1024
        elif model == "vit_b_16":
            self.backbone = ViTWrapper(weights=None)
            proj_input_dim, output_dim = 768, 512
        else:
            raise ValueError(f"Unsupported model: {model}")

        self.projection_head = SimCLRProjectionHead(proj_input_dim, proj_input_dim, output_dim)

        # momentum encoder
        self.backbone_momentum = copy.deepcopy(self.backbone)
        self.projection_head_momentum = copy.deepcopy(self.projection_head)
        deactivate_requires_grad(self.backbone_momentum)
        deactivate_requires_grad(self.projection_head_momentum)

    def _vit_project(self, backbone, head, x):
        if head.training:
            # train mode (HP/src/pretrain_engine.py:603 model.train()): the backbone forward is differentiable on the
            # HIP path (hcir.vit_train); the lightly head is torch modules as in the reference - BatchNorm1d with
            # batch statistics and running-stat updates, autograd by torch (B x 768 x 768: 0.1 % of the step)
            return head(backbone.forward_cls(x))
        _, cls16 = backbone.forward_cls(x, l2_normalize=False, want_f16=True)
        return head.forward_hip(cls16)

    def forward(self, x):
        if "vit" in self.model:
            return self._vit_project(self.backbone, self.projection_head, x)
        x = self.backbone(x).flatten(start_dim=1)
        return self.projection_head(x)

    def forward_views(self, views):
        """[self(v) for v in views] — what the step computes at HP/src/pretrain_engine.py:683-690 (negatives,
        positives, anchors) — with ONE differentiable backbone pass over the concatenated views: the ViT has no batch
        coupling (LayerNorm only), so every row comes out as in a pass of its own, while the GEMM / TN-GEMM tile
        counts fill whole rounds of the 256 CUs and the launch count drops to a third.  The projection head, whose
        BatchNorm1d takes BATCH statistics, still runs once per view, in order (running statistics as in the
        reference)."""
        if "vit" in self.model and self.projection_head.training:
            sizes = [int(v.shape[0]) for v in views]
            cls = self.backbone.forward_cls(torch.cat(list(views)))
            return [self.projection_head(c) for c in cls.split(sizes)]
        return [self.forward(v) for v in views]

    @torch.no_grad()
    def forward_momentum(self, x):
        if "vit" in self.model:
            return self._vit_project(self.backbone_momentum, self.projection_head_momentum, x)
        x = self.backbone_momentum(x).flatten(start_dim=1)
        return self.projection_head_momentum(x)

    def extract_features(self, x):
        if "vit" in self.model:
            return self.backbone.forward_cls(x)
        return self.backbone(x).flatten(start_dim=1)

    @torch.no_grad()
    def extract_features_ema(self, x):
        if "vit" in self.model:
            return self.backbone_momentum.forward_cls(x)
        return self.backbone_momentum(x).flatten(start_dim=1)
