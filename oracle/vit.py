"""oracle.vit — torch-CPU fp32 restatement of the reference's backbone forwards.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional: every function takes
a state_dict keyed by the REFERENCE's parameter names plus an input tensor.

PARITY UNPINNED for the third-party halves: the arithmetic of these paths lives in
torchvision (vit_b_16, resnet50), timm (VisionTransformer base, Mlp) and lightly
(SimCLRProjectionHead) — all un-pinned in requirements.txt:1-8 and absent from this
image, and the reference ships no tests or golden vectors.  What
IS executable here and is used verbatim: torch's own F.multi_head_attention_forward,
F.layer_norm, F.conv2d, F.gelu, F.linear, F.batch_norm — the ATen ops those
libraries' modules call.  Only the few lines of composition are restated, each citing
the reference line (or, for third-party code).

  vitwrapper_forward     HP/src/main_backbone.py:539-563  (+ torchvision Encoder /
                         EncoderBlock composition)
  models_vit_forward_features  HP/src/models_vit.py:227-241 with Block :147-150,
                         Attention :69-80
  resnet_trunk_forward   torchvision resnet (v1.5) children()[:-1], HP/src/main_backbone.py:577-578
  projection_head_forward lightly SimCLRProjectionHead, HP/src/main_backbone.py:589
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

POS_EMBED_MULT = 2.0  # wrapper adds pos, then torchvision's Encoder adds the same tensor again


def _tv_encoder_block(sd, pre, x, num_heads):
    """torchvision EncoderBlock.forward:
        x = ln_1(inp); x,_ = self_attention(x,x,x,need_weights=False); x = x + inp
        y = ln_2(x); y = mlp(y); return x + y
    """
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"], 1e-6)
    # nn.MultiheadAttention(batch_first=True) transposes to (T, B, D) and calls this:
    hq = h.transpose(0, 1)
    a, _ = F.multi_head_attention_forward(
        hq, hq, hq, d, num_heads,
        sd[pre + "self_attention.in_proj_weight"], sd[pre + "self_attention.in_proj_bias"],
        None, None, False, 0.0,
        sd[pre + "self_attention.out_proj.weight"], sd[pre + "self_attention.out_proj.bias"],
        training=False, need_weights=False)
    x = x + a.transpose(0, 1)
    y = F.layer_norm(x, (d,), sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"], 1e-6)
    y = F.linear(y, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])
    y = F.gelu(y)  # nn.GELU() default: exact erf
    y = F.linear(y, sd[pre + "mlp.3.weight"], sd[pre + "mlp.3.bias"])
    return x + y


@torch.no_grad()
def vitwrapper_forward(sd, x, prefix="", num_heads=12):
    """ViTWrapper.forward (HP/src/main_backbone.py:539-563) -> (cls_token, pooled_patches)."""
    p = prefix
    n = x.shape[0]
    x = F.conv2d(x, sd[p + "conv_proj.weight"], sd[p + "conv_proj.bias"], stride=sd[p + "conv_proj.weight"].shape[-1])  # :543
    x = x.flatten(2).transpose(1, 2)                               # :544
    x = torch.cat((sd[p + "cls_token"].expand(n, -1, -1), x), 1)   # :547-548
    x = x + sd[p + "pos_embedding"][:, : x.size(1)]                # :551
    # torchvision Encoder.forward: input = input + self.pos_embedding; ln(layers(dropout(input)))  :554
    x = x + sd[p + "encoder.pos_embedding"]
    i = 0
    while f"{p}encoder.layers.encoder_layer_{i}.ln_1.weight" in sd:
        x = _tv_encoder_block(sd, f"{p}encoder.layers.encoder_layer_{i}.", x, num_heads)
        i += 1
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd[p + "encoder.ln.weight"], sd[p + "encoder.ln.bias"], 1e-6)
    return x[:, 0], x[:, 1:].mean(dim=1)                           # :557-561


@torch.no_grad()
def models_vit_forward_features(sd, x, num_heads=12, prefix="", eps=1e-6):
    """VisionTransformer.forward_features (HP/src/models_vit.py:227-241): patch_embed ->
    cat cls -> + pos_embed -> blocks -> NO final norm.  Block (:147-150):
    x = x + ls1(attn(norm1(x))); x = x + ls2(mlp(norm2(x))).  Attention (:69-80)."""
    p = prefix
    b = x.shape[0]
    w = sd[p + "patch_embed.proj.weight"]
    x = F.conv2d(x, w, sd[p + "patch_embed.proj.bias"], stride=w.shape[-1])   # :48
    x = x.flatten(2).transpose(1, 2)                                           # :50
    x = torch.cat((sd[p + "cls_token"].expand(b, -1, -1), x), dim=1)           # :231-232
    x = x + sd[p + "pos_embed"]                                                # :233
    i = 0
    while f"{p}blocks.{i}.norm1.weight" in sd:
        q = f"{p}blocks.{i}."
        d = x.shape[-1]
        hd = d // num_heads
        h = F.layer_norm(x.float(), (d,), sd[q + "norm1.weight"], sd[q + "norm1.bias"], eps)   # :23-27
        B, N, C = h.shape
        qkv = F.linear(h, sd[q + "attn.qkv.weight"], sd.get(q + "attn.qkv.bias"))
        qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)      # :70
        qq, kk, vv = qkv.unbind(0)
        attn = (qq * hd ** -0.5) @ kk.transpose(-2, -1)                        # :73
        attn = attn - attn.max(-1)[0].unsqueeze(-1)                            # :74
        attn = attn.softmax(dim=-1)                                            # :75
        a = (attn @ vv).transpose(1, 2).reshape(B, N, C)                       # :78
        a = F.linear(a, sd[q + "attn.proj.weight"], sd[q + "attn.proj.bias"])  # :79
        if q + "ls1.gamma" in sd:
            a = a.float() * sd[q + "ls1.gamma"].float()                        # :125
        x = x + a
        h = F.layer_norm(x.float(), (d,), sd[q + "norm2.weight"], sd[q + "norm2.bias"], eps)
        h = F.linear(h, sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"])     # timm Mlp: fc1 -> GELU -> fc2
        h = F.gelu(h)
        h = F.linear(h, sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"])
        if q + "ls2.gamma" in sd:
            h = h.float() * sd[q + "ls2.gamma"].float()
        x = x + h
        i += 1
    return x


def _bn(sd, pre, x):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"],
                        sd[pre + "bias"], False, 0.1, 1e-5)


@torch.no_grad()
def resnet_trunk_forward(sd, x, prefix=""):
    """torchvision resnet18/50 children()[:-1] in eval mode; keys as nn.Sequential
    indexes them: 0 conv1, 1 bn1, 4..7 layer1..4."""
    p = prefix
    x = F.conv2d(x, sd[p + "0.weight"], None, 2, 3)
    x = F.relu(_bn(sd, p + "1.", x))
    x = F.max_pool2d(x, 3, 2, 1)
    for li in (4, 5, 6, 7):
        bi = 0
        while f"{p}{li}.{bi}.conv1.weight" in sd:
            q = f"{p}{li}.{bi}."
            bottleneck = q + "conv3.weight" in sd
            stride = 2 if (li > 4 and bi == 0) else 1
            idt = x
            if q + "downsample.0.weight" in sd:
                idt = _bn(sd, q + "downsample.1.", F.conv2d(x, sd[q + "downsample.0.weight"], None, stride))
            if bottleneck:
                o = F.relu(_bn(sd, q + "bn1.", F.conv2d(x, sd[q + "conv1.weight"])))
                o = F.relu(_bn(sd, q + "bn2.", F.conv2d(o, sd[q + "conv2.weight"], None, stride, 1)))
                o = _bn(sd, q + "bn3.", F.conv2d(o, sd[q + "conv3.weight"]))
            else:
                o = F.relu(_bn(sd, q + "bn1.", F.conv2d(x, sd[q + "conv1.weight"], None, stride, 1)))
                o = _bn(sd, q + "bn2.", F.conv2d(o, sd[q + "conv2.weight"], None, 1, 1))
            x = F.relu(o + idt)
            bi += 1
    return F.adaptive_avg_pool2d(x, 1)


@torch.no_grad()
def projection_head_forward(sd, x, prefix="projection_head."):
    """lightly SimCLRProjectionHead in eval mode: Linear(no bias) -> BN1d -> ReLU -> Linear -> BN1d."""
    p = prefix + "layers."
    x = F.linear(x, sd[p + "0.weight"])
    x = F.relu(F.batch_norm(x, sd[p + "1.running_mean"], sd[p + "1.running_var"], sd[p + "1.weight"],
                            sd[p + "1.bias"], False, 0.1, 1e-5))
    x = F.linear(x, sd[p + "3.weight"])
    return F.batch_norm(x, sd[p + "4.running_mean"], sd[p + "4.running_var"], sd[p + "4.weight"],
                        sd[p + "4.bias"], False, 0.1, 1e-5)


@torch.no_grad()
def sham2_extract_features(sd, x, model="vit_b_16", prefix="backbone."):
    """SHAM2.extract_features (HP/src/main_backbone.py:624-629)."""
    if "vit" in model:
        return vitwrapper_forward(sd, x, prefix)[0]
    return resnet_trunk_forward(sd, x, prefix).flatten(start_dim=1)


@torch.no_grad()
def classifier_embed(sd, x, model="vit_b_16"):
    """Classifier.extracting_features body (HP/src/classification_engine.py:49-50)."""
    return F.normalize(sham2_extract_features(sd, x, model), dim=1)
