"""CPU tests: the oracle is pinned before anything is checked against it.

Pins (SURVEY.md §8c — the reference ships no tests or golden vectors of its own):
  * tests/golden/ntxent_ref.npz   outputs of the reference's in-tree NTXentLoss class
  * tests/golden/knn_sklearn.npz  outputs of scikit-learn 1.7.2 driven as the reference drives it
  * tests/golden/normalize.npz    torch.nn.functional.normalize
  * live scikit-learn / torch.nn modules in this process (same image on the GPU box)
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import knn as oknn
from oracle import ntxent as ont
from oracle import vit as ovit
from oracle import transform as otf


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


# ---------------------------------------------------------------- NT-Xent
def test_ntxent_restatements_match_reference_goldens(golden_dir):
    z = _npz(golden_dir, "ntxent_ref.npz")
    for i in range(int(z["n"])):
        z0, z1 = torch.from_numpy(z[f"z0_{i}"]), torch.from_numpy(z[f"z1_{i}"])
        t, want = float(z[f"t_{i}"]), float(z[f"loss_{i}"])
        assert abs(float(ont.ntxent_dualview(z0, z1, t)) - want) <= 1e-6 * max(1.0, abs(want))
        assert abs(float(ont.ntxent_lightly(z0, z1, t)) - want) <= 2e-6 * max(1.0, abs(want))
        assert abs(float(ont.ntxent_f64(z0, z1, t)[0]) - want) <= 2e-6 * max(1.0, abs(want))
    assert abs(float(z["loss_0"]) - 2.593191623687744) < 1e-6   # SURVEY.md Appendix B anchor


def test_ntxent_temperature_error():
    with pytest.raises(ValueError):
        ont.ntxent_lightly(torch.randn(4, 8), torch.randn(4, 8), 0.0)


# ---------------------------------------------------------------- kNN
def test_knn_oracle_matches_sklearn_goldens(golden_dir):
    z = _npz(golden_dir, "knn_sklearn.npz")
    for i in range(int(z["n"])):
        q, g, k = z[f"q_{i}"], z[f"g_{i}"], int(z[f"k_{i}"])
        for mode in (oknn.MODE_CHAIN32, oknn.MODE_F64):
            val, idx = oknn.cosine_topk(q, g, k, qn=oknn.row_invnorm(q, 1e-30), gn=oknn.row_invnorm(g, 1e-30),
                                        mode=mode)
            np.testing.assert_array_equal(idx, z[f"idx_{i}"])
            np.testing.assert_allclose(np.clip(1.0 - val, 0, 2), z[f"dist_{i}"], atol=5e-7, rtol=0)
        d64, i64 = oknn.sklearn_cosine_kneighbors_np(q, g, k)
        np.testing.assert_array_equal(i64, z[f"idx_{i}"])
        np.testing.assert_allclose(d64, z[f"dist_{i}"], atol=5e-7, rtol=0)
        # uniform-weight vote, smallest label on ties
        pred = oknn.knn_vote(idx, z[f"labels_{i}"])
        np.testing.assert_array_equal(pred, z[f"pred_{i}"])
        # retrieval path on un-normalised embeddings (cosine_similarity + argsort[::-1])
        gu, qu = g * z[f"ret_gscale_{i}"], z[f"ret_q_{i}"]
        sim, ridx = oknn.retrieve_similar_np(qu, gu, k)
        np.testing.assert_array_equal(ridx, z[f"ret_idx_{i}"])
        np.testing.assert_allclose(sim, z[f"ret_sim_{i}"], atol=5e-7, rtol=0)


def test_knn_oracle_vs_live_sklearn():
    from sklearn.neighbors import KNeighborsClassifier
    rng = np.random.default_rng(3)
    q = rng.standard_normal((40, 256), dtype=np.float32)
    g = rng.standard_normal((2000, 256), dtype=np.float32)
    dist, idx = KNeighborsClassifier(n_neighbors=7, metric="cosine").fit(g, np.zeros(2000)).kneighbors(q)
    val, oi = oknn.cosine_topk(q, g, 7, qn=oknn.row_invnorm(q, 1e-30), gn=oknn.row_invnorm(g, 1e-30))
    np.testing.assert_array_equal(oi, idx)
    np.testing.assert_allclose(1.0 - val, dist, atol=1e-6, rtol=0)
    with pytest.raises(ValueError):  # the error behaviour libhcir's wrapper mirrors (k > n_samples_fit)
        KNeighborsClassifier(n_neighbors=11, metric="cosine").fit(g[:10], np.zeros(10)).kneighbors(q)


def test_tie_break_is_documented_deviation(golden_dir):
    """Reference paths disagree on exact ties (sklearn: partition order; argsort[::-1]: highest
    index first).  The build's rule: score desc, index asc."""
    z = _npz(golden_dir, "knn_sklearn.npz")
    val, idx = oknn.cosine_topk(z["tie_q"], z["tie_g"], 3)
    assert list(idx[0]) == [0, 1, 2]
    # both reference paths return the same SET in an order of their sort's own making
    # (np.argsort()[::-1] is not stable; sklearn's argpartition order is arbitrary)
    assert sorted(z["tie_argsort_idx"]) == [0, 1, 2]
    assert sorted(z["tie_sklearn_idx"][0]) == [0, 1, 2]


def test_chain32_mode_is_a_single_fmaf_chain():
    """The fp32 oracle mode must equal an explicit python restatement of the documented k-order."""
    rng = np.random.default_rng(0)
    d = 72
    g, q = rng.standard_normal(d, dtype=np.float32), rng.standard_normal(d, dtype=np.float32)
    acc = np.float32(0)
    order = []
    for c in range((d + 31) // 32):
        for cc in range(4):
            for e in range(4):
                order += [32 * c + 8 * cc + e, 32 * c + 8 * cc + e + 4]
    for k in order:
        if k < d:
            acc = np.float32(np.float64(g[k]) * np.float64(q[k]) + np.float64(acc))  # exact product + one rounding
    got = oknn.scores(q[None], g[None])[0, 0]
    assert got == acc


def test_neg_sampler_and_merge_restatements():
    rng = np.random.default_rng(1)
    emb = rng.standard_normal((32, 64), dtype=np.float32)
    idx = oknn.neg_sampler_static_np(emb, 7)
    e = torch.from_numpy(emb)
    en = e / torch.norm(e, dim=1, keepdim=True).clamp(min=1e-8)      # HP/src/neg_sampling.py:35-45
    ref = torch.sort(en @ en.t(), dim=1, descending=True)[1][:, 6]
    np.testing.assert_array_equal(idx, ref.numpy())
    with pytest.raises(ValueError):
        oknn.neg_sampler_static_np(emb, 33)
    vals = -np.sort(-rng.standard_normal((3, 5, 4)).astype(np.float32), axis=2)
    ids = np.arange(60, dtype=np.int64).reshape(3, 5, 4)
    mv, mi = oknn.topk_merge(vals, ids, 4)
    flat_v, flat_i = vals.transpose(1, 0, 2).reshape(5, -1), ids.transpose(1, 0, 2).reshape(5, -1)
    order = np.argsort(-flat_v, axis=1, kind="stable")[:, :4]
    np.testing.assert_array_equal(mv, np.take_along_axis(flat_v, order, 1))


# ---------------------------------------------------------------- normalize / transform
def test_normalize_golden(golden_dir):
    z = _npz(golden_dir, "normalize.npz")
    x = z["x"]
    inv = oknn.row_invnorm(x, 1e-12)
    np.testing.assert_allclose(x * inv[:, None], z["y"], atol=2e-7, rtol=0)
    assert np.all(z["y"][7] == 0)


def test_knn_transform_on_asset_windows(golden_dir):
    z = _npz(golden_dir, "asset_windows.npz")
    win = z["windows"]
    assert win.shape == (4, 224, 224, 3) and win.dtype == np.uint8
    x = otf.window_to_tensor(win[0])
    assert x.shape == (3, 224, 224) and x.dtype == np.float32
    # ToTensor + Normalize, element by element
    ref = (win[0].astype(np.float64) / 255.0 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225])
    np.testing.assert_allclose(x, ref.transpose(2, 0, 1), atol=1e-6)
    from PIL import Image
    full = np.zeros(tuple(z["full_shape"]), dtype=np.uint8)
    h, w = full.shape[:2]
    top, left = int(round((h - 224) / 2.0)), int(round((w - 224) / 2.0))
    full[top:top + 224, left:left + 224] = win[1]
    np.testing.assert_array_equal(otf.knn_transform(Image.fromarray(full)), otf.window_to_tensor(win[1]))


# ---------------------------------------------------------------- backbones (structural pins)
def test_vit_oracle_equals_torch_modules_and_state_dict_layout():
    """The functional oracle must agree with the torch.nn modules torchvision composes, and the
    containers must expose the reference's state-dict keys / shapes."""
    from hcir.main_backbone import SHAM2
    torch.manual_seed(0)
    m = SHAM2("vit_b_16").eval()
    sd = m.state_dict()
    expect = {
        "backbone.conv_proj.weight": (768, 3, 16, 16), "backbone.encoder.pos_embedding": (1, 197, 768),
        "backbone.cls_token": (1, 1, 768), "backbone.pos_embedding": (1, 197, 768),
        "backbone.encoder.layers.encoder_layer_11.self_attention.in_proj_weight": (2304, 768),
        "backbone.encoder.layers.encoder_layer_0.self_attention.out_proj.bias": (768,),
        "backbone.encoder.layers.encoder_layer_3.mlp.0.weight": (3072, 768),
        "backbone.encoder.layers.encoder_layer_3.mlp.3.weight": (768, 3072),
        "backbone.encoder.ln.weight": (768,), "projection_head.layers.0.weight": (768, 768),
        "projection_head.layers.3.weight": (512, 768), "projection_head.layers.4.running_var": (512,),
        "backbone_momentum.encoder.ln.bias": (768,),
    }
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    assert "projection_head.layers.0.bias" not in sd
    assert all(not p.requires_grad for p in m.backbone_momentum.parameters())
    blk = m.backbone.encoder.layers[2]
    h = torch.randn(2, 197, 768)
    with torch.no_grad():
        x = blk.ln_1(h)
        a, _ = blk.self_attention(x, x, x, need_weights=False)
        y = a + h
        ref = y + blk.mlp(blk.ln_2(y))
    got = ovit._tv_encoder_block(sd, "backbone.encoder.layers.encoder_layer_2.", h, 12)
    assert (ref - got).abs().max() < 5e-6


def test_double_positional_add_is_reproduced():
    from hcir.main_backbone import POS_EMBED_MULT, SHAM2
    assert POS_EMBED_MULT == ovit.POS_EMBED_MULT == 2.0
    torch.manual_seed(1)
    m = SHAM2("vit_b_16").eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(1, 3, 224, 224)
    a = ovit.vitwrapper_forward(sd, x, "backbone.")[0]
    sd2 = dict(sd)
    sd2["backbone.pos_embedding"] = sd["backbone.pos_embedding"] * 0 + 0.5 * (sd["backbone.pos_embedding"] * 2)
    b = ovit.vitwrapper_forward(sd2, x, "backbone.")[0]
    assert torch.allclose(a, b)
    sd3 = dict(sd)
    sd3["backbone.pos_embedding"] = torch.zeros_like(sd["backbone.pos_embedding"])
    c = ovit.vitwrapper_forward(sd3, x, "backbone.")[0]
    assert not torch.allclose(a, c, atol=1e-4)   # the wrapper-side add matters


def test_resnet_oracle_equals_modules():
    from hcir.main_backbone import SHAM2
    torch.manual_seed(2)
    for name, dim in (("resnet18", 512), ("resnet50", 2048)):
        m = SHAM2(name).eval()
        for mod in m.modules():   # non-trivial BN statistics
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
        x = torch.randn(2, 3, 224, 224)
        with torch.no_grad():
            ref = m.extract_features(x)
        got = ovit.sham2_extract_features(m.state_dict(), x, name)
        assert ref.shape == (2, dim)
        assert (ref - got).abs().max() <= 1e-4 * ref.abs().max()
    sd = m.state_dict()
    assert tuple(sd["backbone.0.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["backbone.7.2.conv3.weight"].shape) == (2048, 512, 1, 1)


def test_models_vit_oracle_and_keys():
    from hcir.models_vit import vit_base_patch16, vit_large_patch16
    m = vit_base_patch16(drop_path_rate=0.1, global_pool=True, init_values=None)
    sd = m.state_dict()
    for k in ("cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.bias",
              "fc_norm.weight", "head.weight"):
        assert k in sd, k
    assert "norm.weight" not in sd and tuple(sd["pos_embed"].shape) == (1, 197, 768)
    assert not m.pos_embed.requires_grad
    with pytest.raises(KeyError):
        vit_base_patch16(drop_path_rate=0.1, global_pool=True)   # init_values is required, as in the reference
    out = ovit.models_vit_forward_features({k: v for k, v in sd.items()}, torch.randn(1, 3, 224, 224))
    assert out.shape == (1, 197, 768)
    big = vit_large_patch16(drop_path_rate=0.0, global_pool=True, init_values=1e-5)
    assert "blocks.23.ls1.gamma" in big.state_dict()


# ---------------------------------------------------------------- independent pins built from torch's own transformer
def _tel_from_tv(sd, pre, d=768, heads=12, mlp=3072):
    """torch.nn.TransformerEncoderLayer(norm_first=True, gelu, eps 1e-6) IS torch's own pre-LN block:
    x = x + SA(LN1(x)); x = x + FF(LN2(x)) — the same structure as torchvision's EncoderBlock, composed
    by torch, not by this build.  Map a torchvision-keyed block into it."""
    tel = torch.nn.TransformerEncoderLayer(d, heads, mlp, dropout=0.0, activation="gelu", batch_first=True,
                                           norm_first=True, layer_norm_eps=1e-6).eval()
    with torch.no_grad():
        tel.self_attn.in_proj_weight.copy_(sd[pre + "self_attention.in_proj_weight"])
        tel.self_attn.in_proj_bias.copy_(sd[pre + "self_attention.in_proj_bias"])
        tel.self_attn.out_proj.weight.copy_(sd[pre + "self_attention.out_proj.weight"])
        tel.self_attn.out_proj.bias.copy_(sd[pre + "self_attention.out_proj.bias"])
        tel.norm1.weight.copy_(sd[pre + "ln_1.weight"]); tel.norm1.bias.copy_(sd[pre + "ln_1.bias"])
        tel.norm2.weight.copy_(sd[pre + "ln_2.weight"]); tel.norm2.bias.copy_(sd[pre + "ln_2.bias"])
        tel.linear1.weight.copy_(sd[pre + "mlp.0.weight"]); tel.linear1.bias.copy_(sd[pre + "mlp.0.bias"])
        tel.linear2.weight.copy_(sd[pre + "mlp.3.weight"]); tel.linear2.bias.copy_(sd[pre + "mlp.3.bias"])
    return tel


def _randomise(sd, seed):
    """Non-trivial values for every parameter (zero biases / unit LayerNorm gains at init would hide a
    swapped or dropped term)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
        elif k.endswith("ln_1.weight") or k.endswith("ln_2.weight") or k.endswith("ln.weight") or "norm" in k and k.endswith("weight"):
            out[k] = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        elif v.dim() >= 2 and v.shape[-1] > 8:
            out[k] = v.clone() if v.abs().sum() > 0 else 0.02 * torch.randn(v.shape, generator=g)
        else:
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
    return out


def test_vit_oracle_block_equals_torch_transformer_encoder_layer():
    """an INDEPENDENT composition of the same block.  torch's TransformerEncoderLayer /
    TransformerEncoder (slow path forced by grad mode being irrelevant: eval + no nested tensors) against
    oracle.vit._tv_encoder_block and the 12-block stack of vitwrapper_forward."""
    from hcir.main_backbone import SHAM2
    torch.manual_seed(3)
    m = SHAM2("vit_b_16").eval()
    sd = _randomise(m.state_dict(), 5)
    h = torch.randn(2, 197, 768)
    # torch's fused "fast path" is a different kernel; pin against the documented python path too
    for fast in (False, True):
        torch.backends.mha.set_fastpath_enabled(fast)
        try:
            with torch.no_grad():
                pre = "backbone.encoder.layers.encoder_layer_4."
                ref = _tel_from_tv(sd, pre)(h)
                got = ovit._tv_encoder_block(sd, pre, h, 12)
                assert (ref - got).abs().max() <= 2e-5 * ref.abs().max(), fast
                # the whole stack + final LayerNorm: nn.TransformerEncoder(layers, norm)
                layers = [_tel_from_tv(sd, f"backbone.encoder.layers.encoder_layer_{i}.") for i in range(12)]
                enc = torch.nn.TransformerEncoder(layers[0], 12, norm=torch.nn.LayerNorm(768, eps=1e-6),
                                                  enable_nested_tensor=False).eval()
                for i in range(12):
                    enc.layers[i].load_state_dict(layers[i].state_dict())
                enc.norm.weight.copy_(sd["backbone.encoder.ln.weight"])
                enc.norm.bias.copy_(sd["backbone.encoder.ln.bias"])
                x = torch.randn(2, 3, 224, 224)
                # tokens as ViTWrapper builds them (HP/src/main_backbone.py:543-554), pos added twice
                t = F.conv2d(x, sd["backbone.conv_proj.weight"], sd["backbone.conv_proj.bias"], stride=16)
                t = t.flatten(2).transpose(1, 2)
                t = torch.cat((sd["backbone.cls_token"].expand(2, -1, -1), t), 1)
                t = t + sd["backbone.pos_embedding"] + sd["backbone.encoder.pos_embedding"]
                ref_tok = enc(t)
                cls, pooled = ovit.vitwrapper_forward(sd, x, "backbone.")
                assert (ref_tok[:, 0] - cls).abs().max() <= 5e-5 * ref_tok.abs().max(), fast
                assert (ref_tok[:, 1:].mean(1) - pooled).abs().max() <= 5e-5 * ref_tok.abs().max(), fast
        finally:
            torch.backends.mha.set_fastpath_enabled(True)


def test_models_vit_oracle_equals_torch_transformer_encoder():
    """models_vit.forward_features (timm layout: packed qkv Linear, q pre-scaled, max-subtracted softmax,
    no final norm) against torch's TransformerEncoder with the qkv Linear mapped onto in_proj."""
    from hcir.models_vit import vit_base_patch16
    torch.manual_seed(4)
    m = vit_base_patch16(drop_path_rate=0.0, global_pool=True, init_values=None).eval()
    sd = _randomise(m.state_dict(), 6)
    sd["pos_embed"] = 0.02 * torch.randn(sd["pos_embed"].shape, generator=torch.Generator().manual_seed(7))
    layers = []
    for i in range(12):
        q = f"blocks.{i}."
        tel = torch.nn.TransformerEncoderLayer(768, 12, 3072, dropout=0.0, activation="gelu", batch_first=True,
                                               norm_first=True, layer_norm_eps=1e-6).eval()
        with torch.no_grad():
            tel.self_attn.in_proj_weight.copy_(sd[q + "attn.qkv.weight"])
            tel.self_attn.in_proj_bias.copy_(sd[q + "attn.qkv.bias"])
            tel.self_attn.out_proj.weight.copy_(sd[q + "attn.proj.weight"])
            tel.self_attn.out_proj.bias.copy_(sd[q + "attn.proj.bias"])
            tel.norm1.weight.copy_(sd[q + "norm1.weight"]); tel.norm1.bias.copy_(sd[q + "norm1.bias"])
            tel.norm2.weight.copy_(sd[q + "norm2.weight"]); tel.norm2.bias.copy_(sd[q + "norm2.bias"])
            tel.linear1.weight.copy_(sd[q + "mlp.fc1.weight"]); tel.linear1.bias.copy_(sd[q + "mlp.fc1.bias"])
            tel.linear2.weight.copy_(sd[q + "mlp.fc2.weight"]); tel.linear2.bias.copy_(sd[q + "mlp.fc2.bias"])
        layers.append(tel)
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
        t = torch.cat((sd["cls_token"].expand(2, -1, -1), t), 1) + sd["pos_embed"]
        for tel in layers:
            t = tel(t)
        got = ovit.models_vit_forward_features(sd, x)
    assert (t - got).abs().max() <= 5e-5 * t.abs().max()


def test_projection_head_oracle_equals_torch_sequential():
    """lightly's SimCLRProjectionHead is Sequential(Linear(no bias), BN1d, ReLU, Linear(no bias), BN1d)
    (SURVEY.md Appendix A): compose exactly that from torch.nn and compare."""
    g = torch.Generator().manual_seed(8)
    seq = torch.nn.Sequential(torch.nn.Linear(768, 768, bias=False), torch.nn.BatchNorm1d(768), torch.nn.ReLU(),
                              torch.nn.Linear(768, 512, bias=False), torch.nn.BatchNorm1d(512)).eval()
    with torch.no_grad():
        for mod in seq:
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.2, generator=g); mod.running_var.uniform_(0.5, 1.5, generator=g)
                mod.weight.normal_(1, 0.1, generator=g); mod.bias.normal_(0, 0.1, generator=g)
    sd = {"projection_head.layers." + k: v for k, v in seq.state_dict().items()}
    x = torch.randn(16, 768, generator=g)
    with torch.no_grad():
        ref = seq(x)
    got = ovit.projection_head_forward(sd, x)
    assert (ref - got).abs().max() <= 1e-5 * ref.abs().max()


def test_blocked_oracle_scan_equals_scalar_chain():
    """The AVX2-blocked evaluation (used so that 880 x 1M x 768 is a seconds-long check) must equal the
    scalar chain bit for bit, values and indices, including planted ties and ragged d."""
    rng = np.random.default_rng(0)
    for nq, ng, d, k in ((70, 5000, 72, 7), (3, 40000, 768, 10), (130, 9000, 100, 50), (65, 4099, 33, 16)):
        q = rng.standard_normal((nq, d), dtype=np.float32)
        g = rng.standard_normal((ng, d), dtype=np.float32)
        g[ng // 2] = g[3]
        g[ng - 1] = g[3]
        q[0] = g[3]
        qn, gn = oknn.row_invnorm(q, 1e-12), oknn.row_invnorm(g, 1e-12)
        v1, i1 = oknn.cosine_topk(q, g, k, qn=qn, gn=gn, idx_base=5)
        v2, i2 = oknn.cosine_topk(q, g, k, qn=qn, gn=gn, idx_base=5, mode=oknn.MODE_CHAIN32_SCALAR)
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(v1, v2)
        v1, i1 = oknn.cosine_topk(q, g, k)
        v2, i2 = oknn.cosine_topk(q, g, k, mode=oknn.MODE_CHAIN32_SCALAR)
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(v1, v2)
        assert list(i1[0, :3]) == [3, ng // 2, ng - 1]
