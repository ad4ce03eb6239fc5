"""Parameter containers with torchvision's vit_b_16 state-dict layout.

torchvision is not a dependency of this build (and is absent from the image); what
pins the architecture is the reference's checkpoints, i.e. torchvision's state-dict
key names and shapes.  These modules hold parameters
under exactly those names so `load_state_dict(strict=True)` accepts a reference
checkpoint.  They carry NO arithmetic: forward passes run in hcir.vit_engine (HIP).

Initialisers follow torchvision.models.vision_transformer (public source):
  conv_proj: trunc_normal(std=sqrt(1/fan_in)), zero bias; class_token zeros;
  pos_embedding normal(std=0.02); MLP linears xavier_uniform + normal(std=1e-6) bias;
  nn.MultiheadAttention / nn.LayerNorm defaults.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
from torch import nn


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - containers only
        raise RuntimeError(f"{type(self).__name__} is a parameter container; compute runs in "
                           "hcir.vit_engine (HIP)")


class TVMLPBlock(nn.Sequential):
    """torchvision MLPBlock key layout: 0 = Linear(in, mlp), 3 = Linear(mlp, in)."""

    def __init__(self, in_dim: int, mlp_dim: int):
        super().__init__(nn.Linear(in_dim, mlp_dim), nn.GELU(), nn.Dropout(0.0),
                         nn.Linear(mlp_dim, in_dim), nn.Dropout(0.0))
        for m in (self[0], self[3]):
            nn.init.xavier_uniform_(m.weight)
            nn.init.normal_(m.bias, std=1e-6)


class TVEncoderBlock(_NoForward):
    def __init__(self, num_heads: int, hidden_dim: int, mlp_dim: int):
        super().__init__()
        self.num_heads = num_heads
        self.ln_1 = nn.LayerNorm(hidden_dim, eps=1e-6)
        self.self_attention = nn.MultiheadAttention(hidden_dim, num_heads, dropout=0.0, batch_first=True)
        self.dropout = nn.Dropout(0.0)
        self.ln_2 = nn.LayerNorm(hidden_dim, eps=1e-6)
        self.mlp = TVMLPBlock(hidden_dim, mlp_dim)


class TVEncoder(_NoForward):
    def __init__(self, seq_length: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.dropout = nn.Dropout(0.0)
        layers = OrderedDict()
        for i in range(num_layers):
            layers[f"encoder_layer_{i}"] = TVEncoderBlock(num_heads, hidden_dim, mlp_dim)
        self.layers = nn.Sequential(layers)
        self.ln = nn.LayerNorm(hidden_dim, eps=1e-6)


class TVVisionTransformer(_NoForward):
    """vit_b_16(weights=None) minus `heads` (the reference replaces it by Identity)."""

    def __init__(self, image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768,
                 mlp_dim=3072):
        super().__init__()
        self.image_size, self.patch_size = image_size, patch_size
        self.hidden_dim, self.mlp_dim, self.num_heads, self.num_layers = hidden_dim, mlp_dim, num_heads, num_layers
        self.conv_proj = nn.Conv2d(3, hidden_dim, kernel_size=patch_size, stride=patch_size)
        seq_length = (image_size // patch_size) ** 2 + 1
        self.class_token = nn.Parameter(torch.zeros(1, 1, hidden_dim))
        self.encoder = TVEncoder(seq_length, num_layers, num_heads, hidden_dim, mlp_dim)
        self.seq_length = seq_length
        fan_in = self.conv_proj.in_channels * patch_size * patch_size
        nn.init.trunc_normal_(self.conv_proj.weight, std=math.sqrt(1 / fan_in))
        nn.init.zeros_(self.conv_proj.bias)


def vit_b_16(weights=None) -> TVVisionTransformer:
    if weights is not None:
        raise ValueError("pretrained torchvision weights are not available offline; load a checkpoint")
    return TVVisionTransformer()
