"""GPU parity of the remaining §8 rows: NT-Xent (a9), NegSamplerStatic (a10), models_vit (a5),
MAE / SimCLR wrappers, ResNet-50 + top-5 (config C1 shape), Classifier.knn_eval (a6/a7),
HairEncoder.retrieve_similar_images (a8)."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import knn as oknn
from oracle import ntxent as ont
from oracle import transform as otf
from oracle import vit as ovit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(hcir_built):
    assert torch.cuda.is_available()


def _cos_err(a, b):
    return (1.0 - F.cosine_similarity(a.double(), b.double(), dim=-1)).abs().max().item()


# ------------------------------------------------------------------ NT-Xent
def test_ntxent_goldens_fp32(golden_dir):
    """fp32 inputs (exact-fp32 MFMA): |loss - reference| <= 2e-5 (fp32 exp/log + summation order)."""
    from hcir.losses import NTXentLoss
    z = np.load(os.path.join(golden_dir, "ntxent_ref.npz"))
    for i in range(int(z["n"])):
        z0, z1 = torch.from_numpy(z[f"z0_{i}"]).cuda(), torch.from_numpy(z[f"z1_{i}"]).cuda()
        with torch.no_grad():
            loss = NTXentLoss(temperature=float(z[f"t_{i}"]))(z0, z1)
        assert loss.dim() == 0 and loss.is_cuda
        assert abs(loss.item() - float(z[f"loss_{i}"])) <= 2e-5 * max(1.0, abs(float(z[f"loss_{i}"])))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("b,d,t", [(1024, 512, 0.5), (256, 1024, 0.7), (37, 72, 0.07)])
def test_ntxent_config_c3(dtype, tol, b, d, t):
    """BASELINE config C3: B=1024, D=512.  fp16/bf16 inputs as under the reference's autocast;
    tolerance = input rounding of the normalised rows times 1/T, relative to the fp64 oracle."""
    from hcir.losses import ntxent_forward
    g = torch.Generator().manual_seed(b + d)
    z0, z1 = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g)
    z0d, z1d = z0.cuda().to(dtype), z1.cuda().to(dtype)
    loss, lse = ntxent_forward(z0d, z1d, t, want_lse=True)
    ref, ref_lse = ont.ntxent_f64(z0d.float().cpu(), z1d.float().cpu(), t)
    assert abs(loss.item() - ref.item()) <= tol * max(1.0, abs(ref.item())) / min(1.0, t * 2)
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse.numpy(), atol=tol * 10 / min(1.0, t * 2), rtol=0)


def test_ntxent_errors():
    from hcir import HcirError
    from hcir.losses import NTXentLoss
    z = torch.randn(6, 16, device="cuda", requires_grad=True)   # batch not a multiple of 4
    with pytest.raises(HcirError):
        NTXentLoss(0.5)(z, z)
    with pytest.raises(ValueError):
        NTXentLoss(1e-9)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-3), (torch.float16, 6e-3)])
@pytest.mark.parametrize("b,d,t", [(1024, 512, 0.5), (64, 128, 0.07), (36, 72, 0.7)])
def test_ntxent_backward_vs_autograd(dtype, tol, b, d, t):
    """dL/dz0, dL/dz1 of hcir_ntxent_bwd vs torch autograd through the fp64 restatement of the reference's
    loss.  The backward's second product (dU = W.U) runs in fp16 MFMA: tolerance is relative to max|grad|."""
    from hcir.losses import NTXentLoss
    g = torch.Generator().manual_seed(b + d)
    z0, z1 = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g)
    a0 = z0.to(dtype).float().double().requires_grad_(True)   # the values the kernel actually sees
    a1 = z1.to(dtype).float().double().requires_grad_(True)
    (ont.ntxent_dualview(a0, a1, t) * 3.0).backward()
    x0 = z0.cuda().to(dtype).requires_grad_(True)
    x1 = z1.cuda().to(dtype).requires_grad_(True)
    loss = NTXentLoss(t)(x0, x1)
    assert loss.requires_grad
    (loss * 3.0).backward()
    for got, ref in ((x0.grad, a0.grad), (x1.grad, a1.grad)):
        assert got.dtype == dtype and got.shape == ref.shape
        scale = ref.abs().max().item()
        np.testing.assert_allclose(got.float().cpu().numpy(), ref.float().numpy(), atol=tol * scale, rtol=0)


# ------------------------------------------------------------------ backbones
def _randomize(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in list(model.named_parameters()) + list(model.named_buffers()):
            if not p.dtype.is_floating_point:
                continue
            if "running_var" in name:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif "norm" in name and name.endswith("weight") or name.endswith(("ln_1.weight", "ln_2.weight", "ln.weight")):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif "gamma" in name:
                p.copy_(0.5 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() <= 1 or "cls_token" in name or "pos_emb" in name:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))


def test_models_vit_forward_features_vs_oracle():
    from hcir.models_vit import vit_base_patch16
    torch.manual_seed(3)
    m = vit_base_patch16(drop_path_rate=0.1, global_pool=True, init_values=None).eval()
    _randomize(m, 4)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    ref = ovit.models_vit_forward_features(sd, x)
    m = m.cuda()
    with torch.no_grad():
        out = m.forward_features(x.cuda()).cpu()
    assert out.shape == (3, 197, 768)
    assert _cos_err(out[:, 0], ref[:, 0]) <= 1e-3          # what HairEncoder consumes
    assert _cos_err(out.reshape(3 * 197, 768), ref.reshape(3 * 197, 768)) <= 1e-3


def test_models_vit_layerscale():
    from hcir.models_vit import vit_base_patch16
    torch.manual_seed(6)
    m = vit_base_patch16(drop_path_rate=0.0, global_pool=True, init_values=0.1).eval()
    _randomize(m, 7)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(8))
    ref = ovit.models_vit_forward_features(sd, x)
    with torch.no_grad():
        out = m.cuda().forward_features(x.cuda()).cpu()
    assert _cos_err(out[:, 0], ref[:, 0]) <= 1e-3


def test_vit_large_patch14_config_c5_model():
    """BASELINE config C5's model: ViT-L/14 (24 layers, dim 1024, 16 heads, 257 tokens), fp16 MFMA path."""
    from hcir.models_vit import vit_large_patch14
    torch.manual_seed(20)
    m = vit_large_patch14(drop_path_rate=0.0, global_pool=True, init_values=None).eval()
    _randomize(m, 21)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    assert tuple(sd["pos_embed"].shape) == (1, 257, 1024) and tuple(sd["patch_embed.proj.weight"].shape) == (1024, 3, 14, 14)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(22))
    ref = ovit.models_vit_forward_features(sd, x, num_heads=16)
    with torch.no_grad():
        out = m.cuda().forward_features(x.cuda()).cpu()
    assert out.shape == (2, 257, 1024)
    assert _cos_err(out[:, 0], ref[:, 0]) <= 1e-3


def test_vit_large_patch14_fused_path(monkeypatch):
    """ViT-L/14 at batch 4 (M = 1028 token rows): the persistent GEMM kernel with the fp16 residual stream and
    LayerNorm folded into the GEMMs (slices of 1024 / 64 = 16), LayerScale gamma on the residual epilogues."""
    from hcir import vit_engine
    from hcir.models_vit import vit_large_patch14
    monkeypatch.setattr(vit_engine, "DEFAULT_RESID_DTYPE", torch.float16)
    torch.manual_seed(23)
    m = vit_large_patch14(drop_path_rate=0.0, global_pool=True, init_values=0.1).eval()
    _randomize(m, 24)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(25))
    ref = ovit.models_vit_forward_features(sd, x, num_heads=16)
    m = m.cuda()
    outs = {}
    for fuse in (True, False):
        monkeypatch.setattr(vit_engine, "LN_FUSE", fuse)
        with torch.no_grad():
            outs[fuse] = m.forward_features(x.cuda()).cpu()
        print(f"ViT-L/14 LN fold {fuse}: CLS 1-cos = {_cos_err(outs[fuse][:, 0], ref[:, 0]):.2e}")
        assert _cos_err(outs[fuse][:, 0], ref[:, 0]) <= 1e-3
        # every token row, not only the class token
        assert _cos_err(outs[fuse].reshape(-1, 1024), ref.reshape(-1, 1024)) <= 1e-3
    assert _cos_err(outs[True].reshape(-1, 1024), outs[False].reshape(-1, 1024)) <= 1e-4


def test_vit_huge_geometry_head_dim_80():
    """vit_huge_patch14's geometry (embed 1280, 16 heads -> head_dim 80, patch 14; HP/src/models_vit.py:266-270) at a
    depth of 2 (the 32-block model is 632 M parameters): forward_features against the oracle."""
    from functools import partial
    from hcir.models_vit import LayerNorm, VisionTransformer
    torch.manual_seed(7)
    m = VisionTransformer(patch_size=14, embed_dim=1280, depth=2, num_heads=16, mlp_ratio=4, qkv_bias=True,
                          norm_layer=partial(LayerNorm, eps=1e-6), drop_path_rate=0.0, global_pool=True,
                          init_values=None).eval()
    with torch.no_grad():
        m.pos_embed.normal_(0, 0.02)
        for n, p in m.named_parameters():
            if p.dim() == 1 and "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape))
    x = torch.randn(2, 3, 224, 224)
    ref = ovit.models_vit_forward_features({k: v.clone() for k, v in m.state_dict().items()}, x, num_heads=16)
    m = m.cuda()
    with torch.no_grad():
        out = m.forward_features(x.cuda())
    assert out.shape == (2, 257, 1280)
    assert _cos_err(out.float().cpu(), ref) <= 1e-3


def test_mae_extract_features():
    from hcir.backbone import MAE, vit_base_patch16_224
    torch.manual_seed(9)
    m = MAE(vit_base_patch16_224()).eval()
    _randomize(m, 10)
    sd = {k[len("backbone.vit."):]: v.clone() for k, v in m.state_dict().items() if k.startswith("backbone.vit.")}
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(11))
    tok = ovit.models_vit_forward_features(sd, x)
    ref = F.layer_norm(tok, (768,), sd["norm.weight"], sd["norm.bias"], 1e-6)[:, 0]
    with torch.no_grad():
        out = m.cuda().extract_features(x.cuda()).cpu()
    assert _cos_err(out, ref) <= 1e-3


def test_config_c1_resnet50_top5(golden_dir):
    """BASELINE config C1 shape: ResNet-50 embed of 64 crops cut from the sample assets, top-5 over a
    1000 x 2048 gallery.  ResNet runs on PyTorch-ROCm (MIOpen); the scan is hcir_sim_topk.
    Embedding tolerance 1e-3 cosine; top-5 of the GPU embeddings bit-exact vs the oracle scan."""
    from hcir import ops
    from hcir.main_backbone import SHAM2
    torch.manual_seed(42)
    m = SHAM2("resnet50").eval()
    _randomize(m, 12)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    win = np.load(os.path.join(golden_dir, "asset_windows.npz"))["windows"]
    rng = np.random.default_rng(0)
    crops = [otf.window_to_tensor(w) for w in win]
    while len(crops) < 64:   # further deterministic 224^2 windows: shifted/flipped views of the asset windows
        w = win[len(crops) % 4]
        dy, dx = rng.integers(0, 32, 2)
        crops.append(otf.window_to_tensor(np.roll(w, (dy, dx), (0, 1))[:, ::(-1) ** len(crops)]))
    x = torch.from_numpy(np.stack(crops))
    ref = ovit.classifier_embed(sd, x, "resnet50")
    g = F.normalize(torch.randn(1000, 2048, generator=torch.Generator().manual_seed(0)), dim=1)
    with torch.no_grad():
        emb = ops.l2_normalize(m.cuda().extract_features(x.cuda()).contiguous())
        val, idx = ops.sim_topk(emb, g.cuda(), 5)
    assert _cos_err(emb.cpu(), ref) <= 1e-3
    rv, ri = oknn.cosine_topk(emb.cpu().numpy(), g.numpy(), 5)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)


# ------------------------------------------------------------------ engines
class _FixedFeatureModel(torch.nn.Module):
    """Duck-typed model (extract_features) returning rows of a fixed table keyed by the first pixel."""

    def __init__(self, table):
        super().__init__()
        self.table = torch.nn.Parameter(table, requires_grad=False)

    def extract_features(self, x):
        return self.table[x[:, 0, 0, 0].long()]


def test_classifier_knn_eval_vs_sklearn(tmp_path):
    from sklearn.neighbors import KNeighborsClassifier
    from hcir.classification_engine import Classifier
    rng = np.random.default_rng(1)
    ntr, nte, d, ncls = 900, 120, 96, 6
    centers = rng.standard_normal((ncls, d)).astype(np.float32) * 1.5
    ytr, yte = rng.integers(0, ncls, ntr), rng.integers(0, ncls, nte)
    ftr = centers[ytr] + rng.standard_normal((ntr, d)).astype(np.float32)
    fte = centers[yte] + rng.standard_normal((nte, d)).astype(np.float32)
    table = torch.from_numpy(np.concatenate([ftr, fte]))

    def loader(lo, n, ys):
        out = []
        for s in range(0, n, 64):
            e = min(n, s + 64)
            img = torch.zeros(e - s, 3, 2, 2)
            img[:, 0, 0, 0] = torch.arange(lo + s, lo + e).float()
            out.append((img, torch.from_numpy(ys[s:e])))
        return out

    args = types.SimpleNamespace(device="cuda", mode="SHAM", model="vit_b_16", SHAM_mode="embedding",
                                 save_path=str(tmp_path))
    clf = Classifier(_FixedFeatureModel(table), loader(0, ntr, ytr), loader(ntr, nte, yte), args)
    ks = (5, 10, 27, 642)
    clf.knn_eval(ks=ks)
    txt = open(tmp_path / "SHAM_vit_b_16_embedding" / "knn_evaluation_results.txt").read()
    gtr = F.normalize(torch.from_numpy(ftr), dim=1)
    gte = F.normalize(torch.from_numpy(fte), dim=1)
    for k in ks:
        knn = KNeighborsClassifier(n_neighbors=k, metric="cosine").fit(gtr, torch.from_numpy(ytr))
        acc = float((knn.predict(gte) == yte).mean())
        assert f"Results for k={k}\n" in txt
        assert f"Accuracy: {acc:.4f}" in txt.split(f"Results for k={k}\n")[1].split("=" * 50)[0]
    dist, idx = clf.kneighbors(10)
    sd, si = KNeighborsClassifier(n_neighbors=10, metric="cosine").fit(gtr, ytr).kneighbors(gte)
    np.testing.assert_array_equal(idx.cpu().numpy(), si)
    np.testing.assert_allclose(dist.cpu().numpy(), sd, atol=1e-6, rtol=0)
    with pytest.raises(ValueError):     # k > n_samples_fit: sklearn's error, mirrored
        clf.knn_eval(ks=(901,))


def test_neg_sampler_static():
    from hcir.neg_sampling import NegSamplerStatic
    rng = np.random.default_rng(2)
    emb = torch.from_numpy(rng.standard_normal((256, 768)).astype(np.float32) * 2)
    model = types.SimpleNamespace(extract_features_ema=lambda batch: emb.cuda()[batch.long()])
    batch = torch.arange(256).cuda()
    for k in (1, 7, 15):
        got = NegSamplerStatic(model, batch, metric="cosine", k=k).cpu().numpy()
        np.testing.assert_array_equal(got, oknn.neg_sampler_static_np(emb.numpy(), k))
    assert (NegSamplerStatic(model, batch, k=1).cpu().numpy() == np.arange(256)).all()  # self is rank 0
    with pytest.raises(ValueError):
        NegSamplerStatic(model, batch, k=257)
    with pytest.raises(ValueError):
        NegSamplerStatic(model, batch, metric="manhattan", k=3)


def test_hair_encoder_retrieve(golden_dir, tmp_path):
    from hcir.hair_encoder import HairEncoder
    z = np.load(os.path.join(golden_dir, "knn_sklearn.npz"))
    enc = HairEncoder(None, "vit_base_patch16", device="cuda")
    for i in range(int(z["n"])):
        g = z[f"g_{i}"] * z[f"ret_gscale_{i}"]
        paths = [f"img_{j}.png" for j in range(len(g))]
        res = enc.retrieve_similar_images(z[f"ret_q_{i}"], g, paths, top_k=int(z[f"k_{i}"]))
        assert [r["path"] for r in res] == [paths[j] for j in z[f"ret_idx_{i}"]]
        np.testing.assert_allclose([r["similarity"] for r in res], z[f"ret_sim_{i}"], atol=1e-6, rtol=0)
    enc.save_embeddings(g, paths, str(tmp_path))
    assert enc.check_embeddings_exist(str(tmp_path))
    g2, p2 = enc.load_embeddings(str(tmp_path))
    assert np.array_equal(g2, g) and p2 == paths
    x = torch.randn(2, 3, 224, 224)
    feats = enc.extract_features(x.cuda())
    ref = ovit.models_vit_forward_features({k: v.cpu() for k, v in enc.model.state_dict().items()}, x)[:, 0]
    assert _cos_err(feats.cpu(), ref) <= 1e-3


def test_hair_retrieval_flow_end_to_end(golden_dir, tmp_path, capsys):
    """src/hair_retrieval.py main(): extract_dataset_features over an ImageFolder tree -> embeddings.npy +
    image_paths.txt -> encode_single_image -> retrieve_similar_images, against the oracle fed by the
    reference's host transform (Resize(224, bicubic) -> CenterCrop -> ToTensor -> Normalize)."""
    import importlib.util
    from PIL import Image
    from hcir.hair_encoder import resize_shorter_side
    from hcir.transform import knn_transform
    win = np.load(os.path.join(golden_dir, "asset_windows.npz"))["windows"]
    root = tmp_path / "data"
    imgs = []
    rng = np.random.default_rng(0)
    for c in range(3):
        os.makedirs(root / f"class{c}")
        for j in range(4):
            w = win[(c + j) % 4]
            arr = np.roll(w, (17 * c, 29 * j), axis=(0, 1))
            arr = np.pad(arr, ((10 * j, 0), (0, 40 * c), (0, 0)))          # non-square: Resize really resizes
            arr = (arr.astype(np.int32) + rng.integers(0, 20, arr.shape)).clip(0, 255).astype(np.uint8)
            path = root / f"class{c}" / f"img{j}.png"
            Image.fromarray(arr).save(path)
            imgs.append((str(path), arr))
    spec = importlib.util.spec_from_file_location(
        "hair_cli", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 "hair-centric-image-retrieval_amd", "hair_retrieval.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    emb_dir = str(tmp_path / "emb")
    argv = ["--data_path", str(root), "--embed_save_dir", emb_dir, "--batch_size", "5", "--num_workers", "0",
            "--device", "cuda", "--top_k", "4", "--query_image", imgs[6][0]]
    torch.manual_seed(0)
    qpath, results = cli.main(argv)
    emb = np.load(os.path.join(emb_dir, "embeddings.npy"))
    paths = [l.strip() for l in open(os.path.join(emb_dir, "image_paths.txt"))]
    assert emb.shape == (12, 768) and paths == [p for p, _ in imgs]       # ImageFolder order
    # oracle: same weights (seeded ctor), reference host pipeline
    from hcir.hair_encoder import HairEncoder
    torch.manual_seed(0)
    enc = HairEncoder(None, "vit_base_patch16", device="cuda")
    sd = {k: v.cpu() for k, v in enc.model.state_dict().items()}
    x = torch.stack([knn_transform(resize_shorter_side(Image.fromarray(a), 224)) for _, a in imgs])
    ref = ovit.models_vit_forward_features(sd, x)[:, 0]
    assert _cos_err(torch.from_numpy(emb), ref) <= 1e-3
    sim, ridx = oknn.retrieve_similar_np(ref[6].numpy(), ref.numpy(), 4)
    assert results[0]["path"] == imgs[6][0] and abs(results[0]["similarity"] - 1.0) < 1e-3
    np.testing.assert_allclose([r["similarity"] for r in results], sim, atol=2e-3)
    # second run: embeddings exist -> loaded, not re-extracted; --retrieve_only / multiple queries
    out = cli.main(argv[:8] + ["--top_k", "3", "--save_visualization", "--num_queries", "2", "--retrieve_only"])
    assert len(out) == 2 and all(len(r) == 3 and q not in [x["path"] for x in r] for q, r in out)
    assert "Loading existing embeddings" in capsys.readouterr().out
    # the gallery cache follows the array passed in (id() re-use / in-place edits)
    g = emb.copy()
    r1 = enc.retrieve_similar_images(emb[2], g, paths, top_k=2)
    g[:] = g[::-1].copy()                                                  # in-place edit of the same object
    r2 = enc.retrieve_similar_images(emb[2], g, paths, top_k=2)
    assert r1[0]["path"] == paths[2] and r2[0]["path"] == paths[12 - 1 - 2]


def test_knn_cli_end_to_end(tmp_path):
    """The reference's CLI contract on a synthetic folder: same flags, same output file; the sweep
    stops with sklearn's ValueError at the first k larger than the training set (k = 642 is in the
    reference's default sweep, HP/src/classification_engine.py:71)."""
    import importlib.util
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        "knn_cli_e2e", os.path.join(root, "hair-centric-image-retrieval_amd", "knn_classification.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    rng = np.random.default_rng(0)
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    rows = {"train": [], "test": []}
    for split, n in (("train", 48), ("test", 12)):
        for i in range(n):
            cls = i % 3
            arr = rng.integers(0, 255, (240, 250, 3), dtype=np.uint8)
            arr[..., cls] = 255 - arr[..., cls] // 4          # class-dependent colour cast
            name = f"{split}_{i}.png"
            Image.fromarray(arr).save(img_dir / name)
            rows[split].append(f"{name},{cls}")
    for split in rows:
        (tmp_path / f"{split}.csv").write_text("id,class\n" + "\n".join(rows[split]) + "\n")
    args = cli.parse_args(["--mode", "SHAM", "--model", "resnet18", "--eval_type", "knn", "--batch_size", "16",
                           "--num_workers", "0", "--device", "cuda", "--save_path", str(tmp_path / "out"),
                           "--train_annotation", str(tmp_path / "train.csv"),
                           "--test_annotation", str(tmp_path / "test.csv"), "--img_dir", str(img_dir)])
    cli.set_seed(args.seed)
    with pytest.raises(ValueError, match="n_neighbors"):
        cli.main(args)
    txt = (tmp_path / "out" / "SHAM_resnet18_embedding" / "knn_evaluation_results.txt").read_text()
    for k in (5, 10, 20, 27, 30, 40):
        assert f"Results for k={k}\n" in txt
    assert "Results for k=642" not in txt and "Confusion Matrix:" in txt


def test_index_flat_l2(tmp_path):
    """faiss.IndexFlatL2 semantics (squared L2, ascending, -1 / +inf padding) on hcir_sim_topk, checked
    against float64 numpy; unit-norm rows (the reference's use) and general rows."""
    from hcir import index as hidx
    rng = np.random.default_rng(4)
    for normalise, d in ((True, 768), (False, 100)):
        g = rng.standard_normal((3000, d)).astype(np.float32) * (1.0 if normalise else 3.0)
        q = rng.standard_normal((17, d)).astype(np.float32)
        if normalise:
            hidx.normalize_L2(g)
            hidx.normalize_L2(q)
            assert np.allclose(np.linalg.norm(g, axis=1), 1.0, atol=1e-6)
        ix = hidx.IndexFlatL2(d)
        ix.add(g[:1000])
        ix.add(g[1000:])
        assert ix.ntotal == 3000 and ix.is_trained
        D, I = ix.search(q, 10)
        ref = ((q[:, None, :].astype(np.float64) - g[None].astype(np.float64)) ** 2).sum(-1)
        order = np.argsort(ref, axis=1, kind="stable")[:, :11]
        rd = np.take_along_axis(ref, order, 1)
        tol = 2e-5 * max(1.0, rd.max())
        np.testing.assert_allclose(D, rd[:, :10], atol=tol, rtol=0)
        safe = (rd[:, 1:] - rd[:, :-1]) > 4 * tol
        ok = np.concatenate([np.ones((17, 1), bool), safe[:, :9]], 1) & safe[:, :10]
        np.testing.assert_array_equal(I[ok], order[:, :10][ok])
        if normalise:      # on unit vectors D = 2 - 2 cos: the ranking the reference relies on
            np.testing.assert_allclose(D, 2 - 2 * np.take_along_axis(q.astype(np.float64) @ g.T.astype(np.float64), I, 1),
                                       atol=1e-5)
    small = hidx.IndexFlatL2(d)
    small.add(g[:7])
    D, I = small.search(q[:2], 10)                      # k > ntotal: padded like faiss
    assert (I[:, 7:] == -1).all() and np.isinf(D[:, 7:]).all() and (I[:, :7] >= 0).all()
    with pytest.raises(ValueError):                     # the scan's k limit (HCIR_TOPK_MAX)
        ix.search(q[:2], 2000)
    hidx.write_index(ix, str(tmp_path / "hair.index"))
    ix2 = hidx.read_index(str(tmp_path / "hair.index"))
    D2, I2 = ix2.search(q, 10)
    D1, I1 = ix.search(q, 10)
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)
    hidx.save_paths(["a.png", "b.png"], str(tmp_path / "paths.pkl"))
    assert hidx.load_paths(str(tmp_path / "paths.pkl")) == ["a.png", "b.png"]


# ------------------------------------------------------------------ knn_transform on the device (a1)
@pytest.mark.parametrize("h,w", [(224, 224), (1024, 1024), (301, 257), (225, 640), (100, 300), (223, 222)])
def test_knn_transform_u8_bit_exact(h, w):
    """hcir_knn_transform_u8 == CenterCrop(224) -> ToTensor -> Normalize element for element (IEEE fp32),
    including odd differences (Python round half to even) and images smaller than the window (zero pad)."""
    from PIL import Image
    from hcir.transform import center_window_u8, knn_transform, knn_transform_u8
    rng = np.random.default_rng(h * 1000 + w)
    imgs = rng.integers(0, 256, size=(3, h, w, 3), dtype=np.uint8)
    got = knn_transform_u8(torch.from_numpy(imgs).cuda()).cpu().numpy()
    for i in range(3):
        pil = Image.fromarray(imgs[i])
        ref = knn_transform(pil).numpy()                      # host form, same contract as the reference's
        np.testing.assert_array_equal(got[i], ref)
        if h >= 224 and w >= 224:
            np.testing.assert_array_equal(ref, otf.knn_transform(pil))      # oracle restatement
        # loaders may ship the window only: same result
        win = center_window_u8(pil)
        assert win.dtype == torch.uint8 and tuple(win.shape) == (224, 224, 3)
        np.testing.assert_array_equal(knn_transform_u8(win.cuda())[0].cpu().numpy(), ref)


def test_knn_transform_u8_goldens_and_errors(golden_dir):
    from hcir import HcirError
    from hcir.transform import knn_transform_u8
    z = np.load(os.path.join(golden_dir, "asset_windows.npz"))
    wins = z["windows"] if "windows" in z else z[z.files[0]]
    assert wins.dtype == np.uint8 and wins.shape[1:] == (224, 224, 3)
    got = knn_transform_u8(torch.from_numpy(wins).cuda()).cpu().numpy()
    for i in range(wins.shape[0]):
        np.testing.assert_array_equal(got[i], otf.window_to_tensor(wins[i]))
    with pytest.raises(HcirError):
        knn_transform_u8(torch.zeros(1, 224, 224, 3, dtype=torch.uint8))          # CPU tensor: no fallback
    with pytest.raises(HcirError):
        knn_transform_u8(torch.zeros(1, 224, 224, 3, device="cuda"))               # not uint8


# ------------------------------------------------------------------ momentum encoder update (pretrain step)
def test_update_momentum_bit_exact():
    """hcir.momentum.update_momentum == lightly's `ema.data = ema.data * m + p.data * (1.0 - m)` bit for bit,
    over every parameter of SHAM2's backbone + projection head in one launch; the engine cache sees the update."""
    from hcir import HcirError
    from hcir.main_backbone import SHAM2
    from hcir.momentum import update_momentum
    torch.manual_seed(50)
    model = SHAM2("vit_b_16").cuda()
    _randomize(model, 51)
    with torch.no_grad():
        for p in model.backbone_momentum.parameters():
            p.add_(0.01 * torch.randn_like(p))
    for m in (0.99, 0.996, 0.5):
        want = [e.data * m + p.data * (1.0 - m)
                for p, e in zip(model.backbone.parameters(), model.backbone_momentum.parameters())]
        vers = [e._version for e in model.backbone_momentum.parameters()]
        update_momentum(model.backbone, model.backbone_momentum, m)
        update_momentum(model.projection_head, model.projection_head_momentum, m)
        torch.cuda.synchronize()
        for e, w, v in zip(model.backbone_momentum.parameters(), want, vers):
            assert torch.equal(e.data, w)
            assert e._version > v
    # the momentum forward runs on the updated weights
    x = torch.randn(2, 3, 224, 224, device="cuda")
    with torch.no_grad():
        a = model.extract_features_ema(x)
        update_momentum(model.backbone, model.backbone_momentum, 0.0)      # ema := online weights
        b = model.extract_features_ema(x)
        c = model.extract_features(x)
    assert not torch.equal(a, b)
    assert _cos_err(b, c) <= 1e-6
    with pytest.raises(HcirError):
        update_momentum(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), 0.9)    # CPU parameters: no fallback


# ------------------------------------------------------------------ PositiveMaskingTransform (pretrain step)
@pytest.mark.parametrize("b,h,w,patch", [(6, 224, 224, 32), (3, 224, 224, 16), (2, 100, 130, 32)])
def test_positive_masking_vs_oracle(b, h, w, patch):
    """hcir_positive_masking == the reference's per-image loop driven by the same random numbers: identical
    masked images and counts.  Images are hair-on-black composites (patch means far from the 0.01 threshold)."""
    from hcir.transform import PositiveMaskingTransform
    rng = np.random.default_rng(b * 100 + patch)
    img = rng.uniform(0.2, 1.0, size=(b, 3, h, w)).astype(np.float32)
    nh, nw = h // patch, w // patch
    for i in range(b):                                  # black out a random ~40 % of the patches
        dead = rng.random((nh, nw)) < 0.4
        for ph in range(nh):
            for pw in range(nw):
                if dead[ph, pw]:
                    img[i, :, ph * patch:(ph + 1) * patch, pw * patch:(pw + 1) * patch] = 0.0
    img[b - 1] = 0.0                                    # an image with no hair at all is returned unchanged
    u = rng.uniform(0.1, 0.2, size=b).astype(np.float32)
    u[0] = 0.0009                                       # int(n_hair * u) == 0: nothing masked
    keys = rng.random((b, nh * nw)).astype(np.float32)
    keys[1, :4] = keys[1, 5]                            # ties between keys resolve by patch index
    t = PositiveMaskingTransform(patch_size=patch, mask_ratio_range=(0.1, 0.2), threshold=0.01)
    got, cnt = t.apply(torch.from_numpy(img).cuda(), torch.from_numpy(u), torch.from_numpy(keys), return_counts=True)
    ref, rcnt = otf.positive_masking(img, u, keys, patch_size=patch, threshold=0.01)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    assert rcnt[0] == 0 and rcnt[b - 1] == 0 and (b <= 2 or rcnt[1:b - 1].min() >= 1)
    # the drawing form: same contract as the reference's __call__ (10-20 % of the hair patches zeroed)
    out = t(torch.from_numpy(img).cuda(), generator=torch.Generator(device="cuda").manual_seed(3))
    zeroed = ((out == 0).all(dim=1) & (torch.from_numpy(img).cuda() != 0).any(dim=1)).float().sum().item()
    assert b <= 2 or zeroed > 0
    with pytest.raises(ValueError):
        t([1, 2, 3])
