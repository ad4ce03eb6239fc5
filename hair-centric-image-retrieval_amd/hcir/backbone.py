"""hcir.backbone — the two baseline wrappers of HP/src/backbone.py that sit on the kNN path,
with the reference's constructor signatures:

  SimCLR(model="resnet18")   HP/src/backbone.py:648-681  (the second definition, which wins)
  MAE(vit)                   HP/src/backbone.py:462-525  (extract_features = encoder CLS)
  vit_base_patch16_224()     timm ctor the CLI passes to MAE (HP/knn_classification.py:146-147)

The other SSL baselines of that file (MSN, DenseCL, BYOL, DINO, SimMIM, DINOv2, SiameseIMViT)
are comparison methods outside the hot path and are not built.
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from . import _tv_resnet, models_vit
from .main_backbone import SimCLRProjectionHead, ViTWrapper
from .vit_engine import EngineCache, VitLayer, VitSpec


class SimCLR(nn.Module):
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

    def forward(self, x):
        if "vit" in self.model:
            # The reference calls .flatten on ViTWrapper's (cls, pooled) TUPLE here and crashes
            # (HP/src/backbone.py:675-681).  Defined as SHAM2 does: project CLS.
            _, cls16 = self.backbone.forward_cls(x, want_f16=True)
            return self.projection_head.forward_hip(cls16)
        return self.projection_head(self.backbone(x).flatten(start_dim=1))

    def extract_features(self, x):
        if "vit" in self.model:
            return self.backbone.forward_cls(x)
        return self.backbone(x).flatten(start_dim=1)


class _TimmViT(models_vit.VisionTransformer):
    """timm vit_base_patch16_224 container: like models_vit but WITH the final `norm`,
    a trainable pos_embed and no fc_norm (timm defaults: global_pool='token')."""

    def __init__(self, **kwargs):
        kwargs.setdefault("init_values", None)
        kwargs.setdefault("drop_path_rate", 0.0)
        super().__init__(global_pool=True, **kwargs)
        d = self.embed_dim
        del self.fc_norm
        self.norm = kwargs["norm_layer"](d)
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, d) * 0.02)
        self.embed_dim = d


def vit_base_patch16_224(pretrained=False, **kwargs):
    if pretrained:
        raise ValueError("pretrained timm weights are not available offline; load a checkpoint")
    return _TimmViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                    norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


class _MaskedViT(nn.Module):
    """lightly MaskedVisionTransformerTIMM(vit=vit) container: keys backbone.vit.*, backbone.mask_token."""

    def __init__(self, vit):
        super().__init__()
        self.vit = vit
        self.mask_token = nn.Parameter(torch.zeros(1, 1, vit.embed_dim))
        self.sequence_length = vit.patch_embed.num_patches + 1


class MAE(nn.Module):
    """MAE(vit): only the encoder half is on the hot path.  extract_features(images) ==
    backbone.encode(images, idx_keep=None)[:, 0] (HP/src/backbone.py:523-525): timm forward
    (patch_embed, cls, + pos_embed, blocks, final norm), CLS row.  The MAE decoder
    (pre-training only) is not built; load checkpoints with strict=False."""

    def __init__(self, vit):
        super().__init__()
        self.mask_ratio = 0.75
        self.patch_size = vit.patch_embed.patch_size[0]
        self.backbone = _MaskedViT(vit)
        self.sequence_length = self.backbone.sequence_length
        self._cache = EngineCache()

    def _spec(self) -> VitSpec:
        vit = self.backbone.vit
        spec = models_vit.VisionTransformer._spec(vit)
        spec.final_ln_w, spec.final_ln_b = vit.norm.weight, vit.norm.bias
        return spec

    def forward(self, images):
        raise NotImplementedError("MAE pre-training (masking + decoder) is outside the retrieval hot path")

    def extract_features(self, images):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("this model family is inference-only on the HIP path (the differentiable ViT is SHAM2 / ViTWrapper.forward_cls, hcir.vit_train): wrap the call in torch.no_grad()")
        eng = self._cache.get(list(self.backbone.vit.parameters()), self._spec, images.device)
        tok = eng.forward_tokens(images, cls_only_last=True)
        return eng.cls_embedding(tok, final_norm=True, l2_normalize=False)
