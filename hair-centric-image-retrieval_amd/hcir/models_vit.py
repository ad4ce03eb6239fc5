"""hcir.models_vit — the reference's MAE/SiameseIM-style ViT (HP/src/models_vit.py ==
src/models/models_vit.py) with the same constructor signatures and timm state-dict key
names, backed by hcir.vit_engine (HIP).

timm is not a dependency: the timm base class the reference subclasses only contributes
parameter containers (patch_embed.proj, cls_token, pos_embed, norm / fc_norm, head) and
`Mlp(fc1, act, fc2)`; those are restated here as containers.  Arithmetic lives in the HIP
engine; `forward_features` follows HP/src/models_vit.py:227-241:
    patch_embed -> cat(cls) -> + pos_embed -> blocks -> NO final norm.
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from ._lib import HcirError
from .vit_engine import EngineCache, VitLayer, VitSpec


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class LayerNorm(nn.LayerNorm):
    """HP/src/models_vit.py:23-27 — LayerNorm forced to fp32 (the HIP kernel computes in fp32)."""


class Mlp(nn.Module):
    """timm.models.layers.Mlp key layout: fc1, act, fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class DropPath(nn.Identity):
    """Stochastic depth: identity in eval, which is the only mode on the HIP path."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., init_values=None,
                 drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0. else nn.Identity()


class VisionTransformer(nn.Module):
    """Vision Transformer with support for global average pooling (HP/src/models_vit.py:193-250).

    Keyword arguments are timm's: img_size, patch_size, in_chans, num_classes, embed_dim, depth,
    num_heads, mlp_ratio, qkv_bias, norm_layer, drop_path_rate; `init_values` is REQUIRED
    (popped, :197) as is `drop_path_rate` (read, :200) — same KeyErrors as the reference.
    """

    def __init__(self, global_pool=False, **kwargs):
        init_values = kwargs.pop('init_values')
        super().__init__()
        drop_path_rate = kwargs['drop_path_rate']
        depth = kwargs['depth']
        embed_dim = kwargs['embed_dim']
        norm_layer = kwargs['norm_layer']
        img_size = kwargs.get('img_size', 224)
        patch_size = kwargs.get('patch_size', 16)
        num_classes = kwargs.get('num_classes', 1000)
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = kwargs['num_heads']
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size,
                                      in_chans=kwargs.get('in_chans', 3), embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_drop = nn.Dropout(p=kwargs.get('drop_rate', 0.0))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=kwargs['num_heads'], mlp_ratio=kwargs['mlp_ratio'],
                  qkv_bias=kwargs['qkv_bias'], init_values=init_values, norm_layer=norm_layer, drop_path=dpr[i])
            for i in range(depth)])
        self.global_pool = global_pool
        if self.global_pool:
            self.fc_norm = norm_layer(embed_dim)   # the base `norm` is deleted (:212-215)
        else:
            self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        if self.global_pool:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim), requires_grad=False)
            self.cls_pos_embed = nn.Parameter(torch.zeros(1, 1, embed_dim), requires_grad=False)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self._eps = getattr(self.blocks[0].norm1, "eps", 1e-6) if depth else 1e-6
        self._cache = EngineCache()

    def _spec(self) -> VitSpec:
        if not self.global_pool:
            # x + pos_embed with pos_embed [1, num_patches, D] against [B, num_patches+1, D] tokens does
            # not broadcast in the reference either (HP/src/models_vit.py:233): undefined behaviour there.
            raise HcirError("forward_features is only defined for global_pool=True models "
                            "(the reference's HairEncoder, src/models/hair_encoder.py:55-59)")
        layers = []
        for blk in self.blocks:
            layers.append(VitLayer(
                ln1_w=blk.norm1.weight, ln1_b=blk.norm1.bias,
                qkv_w=blk.attn.qkv.weight, qkv_b=blk.attn.qkv.bias,
                proj_w=blk.attn.proj.weight, proj_b=blk.attn.proj.bias,
                ln2_w=blk.norm2.weight, ln2_b=blk.norm2.bias,
                fc1_w=blk.mlp.fc1.weight, fc1_b=blk.mlp.fc1.bias,
                fc2_w=blk.mlp.fc2.weight, fc2_b=blk.mlp.fc2.bias,
                ls1=blk.ls1.gamma if isinstance(blk.ls1, LayerScale) else None,
                ls2=blk.ls2.gamma if isinstance(blk.ls2, LayerScale) else None))
        return VitSpec(patch=self.patch_embed.patch_size[0], dim=self.embed_dim, heads=self.num_heads,
                       eps=self._eps, pos_mult=1.0, conv_w=self.patch_embed.proj.weight,
                       conv_b=self.patch_embed.proj.bias, cls=self.cls_token, pos=self.pos_embed, layers=layers)

    def engine(self, device: torch.device):
        return self._cache.get(list(self.parameters()), self._spec, device)

    def forward_features(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("this model family is inference-only on the HIP path (the differentiable ViT is SHAM2 / ViTWrapper.forward_cls, hcir.vit_train): wrap the call in torch.no_grad()")
        return self.engine(x.device).forward_tokens(x).to(torch.float32, copy=True)  # own fp32 copy, as the reference returns


def vit_base_patch16(**kwargs):
    model = VisionTransformer(
        patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
        norm_layer=partial(LayerNorm, eps=1e-6), **kwargs)
    return model


def vit_large_patch16(**kwargs):
    model = VisionTransformer(
        patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
        norm_layer=partial(LayerNorm, eps=1e-6), **kwargs)
    return model


def vit_large_patch14(**kwargs):
    """ViT-L/14 (BASELINE config C5): not in the reference's models_vit.py (it has large/16 and
    huge/14, :259-270); same constructor convention.  257 tokens, 16 heads of 64."""
    model = VisionTransformer(
        patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
        norm_layer=partial(LayerNorm, eps=1e-6), **kwargs)
    return model


def vit_huge_patch14(**kwargs):
    model = VisionTransformer(
        patch_size=14, embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4, qkv_bias=True,
        norm_layer=partial(LayerNorm, eps=1e-6), **kwargs)
    return model
