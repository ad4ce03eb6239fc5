"""CPU tests: the C-ABI library builds, loads and exports every symbol of include/hcir.h;
host-side logic (sharding, vote, CLI, error behaviour) without any GPU compute."""
import importlib.util
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(hcir_built):
    hdr = open(os.path.join(ROOT, "include", "hcir.h")).read()
    names = set(re.findall(r"\b(hcir_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 15
    from hcir import _lib
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    for n in names:
        assert hasattr(hcir_built, n), n
    assert hcir_built.hcir_version() >= 100
    assert hcir_built.hcir_status_string(-1) == b"invalid argument"
    assert hcir_built.hcir_sim_topk_workspace_bytes(64, 1_000_000, 768, 10, 0) > 0
    assert hcir_built.hcir_ntxent_workspace_bytes(1024, 512, 1) > 0


def test_argument_validation_without_gpu(hcir_built):
    L = hcir_built
    # null pointers / bad shapes are rejected before any launch
    assert L.hcir_sim_topk(None, 1, None, 1, 8, 1, 0, None, None, 0, None, None, None, 0, None) == -1
    assert L.hcir_gemm_f16(None, 8, None, 8, None, None, 1, 8, 8, 0, None, 8, None) == -1
    assert L.hcir_attn_fwd(None, 1, 1, 1, 64, 1.0, 1, None, None) == -1
    assert L.hcir_topk_merge(None, None, 1, 1, 1, 1, None, None, None) == -1


def test_no_cpu_fallback():
    from hcir import HcirError, ops
    from hcir.losses import NTXentLoss
    from hcir.main_backbone import SHAM2
    q = torch.randn(4, 16)
    with pytest.raises(HcirError):
        ops.sim_topk(q, q, 2)
    with pytest.raises(HcirError):
        ops.row_invnorm(q, 1e-12)
    with pytest.raises(HcirError), torch.no_grad():
        NTXentLoss(0.5)(q, q)
    with pytest.raises(HcirError), torch.no_grad():
        SHAM2("vit_b_16").eval().extract_features(torch.randn(1, 3, 224, 224))
    with pytest.raises(ValueError):
        NTXentLoss(0.0)
    with pytest.raises(ValueError):
        SHAM2("alexnet")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "hair-centric-image-retrieval_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)
                assert "knn_oracle" not in src or f.endswith((".hip", ".h")), os.path.join(dp, f)


def test_shard_bounds():
    from hcir.dist import shard_bounds
    for n, w in ((1_000_000, 8), (10, 3), (7, 8), (1, 1)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_knn_vote_matches_oracle():
    from hcir.classification_engine import knn_vote
    from oracle import knn as oknn
    rng = np.random.default_rng(0)
    labels = rng.integers(0, 5, 100)
    nbr = rng.integers(0, 100, (30, 6))
    np.testing.assert_array_equal(knn_vote(labels[nbr], 5), oknn.knn_vote(nbr, labels, 5))
    assert knn_vote(np.array([[2, 1, 1, 2]]), 3)[0] == 1   # tie -> smallest label


def test_cli_flags_match_reference():
    spec = importlib.util.spec_from_file_location(
        "knn_cli", os.path.join(ROOT, "hair-centric-image-retrieval_amd", "knn_classification.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    a = cli.parse_args([])
    # defaults of HP/knn_classification.py:47-67
    assert (a.save_path, a.size, a.batch_size, a.mode, a.model, a.device, a.SHAM_mode, a.seed, a.num_workers,
            a.eval_type) == ("classification_output_dir", 224, 32, "simclr_supcon", "resnet18", "cuda",
                             "embedding", 42, 4, None)
    a = cli.parse_args("--mode SHAM --model vit_b_16 --batch_size 256 --device cuda --num_workers 8 "
                       "--eval_type knn --train_annotation a.csv --test_annotation b.csv --img_dir d".split())
    assert a.mode == "SHAM" and a.eval_type == "knn"
    with pytest.raises(SystemExit):
        cli.parse_args(["--mode", "byol"])
    with pytest.raises(SystemExit):
        cli.build_model(cli.parse_args(["--mode", "dinov2"]))


def test_transform_matches_oracle(golden_dir):
    from PIL import Image
    from hcir.transform import knn_transform
    from oracle import transform as otf
    win = np.load(os.path.join(golden_dir, "asset_windows.npz"))["windows"]
    img = Image.fromarray(np.pad(win[2], ((100, 100), (60, 60), (0, 0))))
    np.testing.assert_array_equal(knn_transform(img).numpy(), otf.knn_transform(img))


def test_committed_bench_line_keeps_the_contract():
    """profiles/r2_bench.json is a verbatim bench.py line: it must carry every key of the bench contract
    (metric/config of BASELINE.json, roofline and cpu_baseline objects) with sane values."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r2_bench.json")) as f:
        line = json.load(f)
    with open(os.path.join(root, "BASELINE.json")) as f:
        base = json.load(f)
    assert line["metric"] == base["metric"]
    for key in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "batch_sweep"):
        assert key in line, key
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert abs(line["value"] - line["config"]["query_batch_per_gpu"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1 and r["traffic"] is not None
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    par = line["parity"]
    assert par["max_1mcos"] <= par["tolerance_1mcos"] == 1e-3 and par["top10_index_match_scan"] == 1.0
    assert set(line["batch_sweep"]) >= {"64", "880"} and "vendor_yardstick_tflops" in r


def test_hair_retrieval_cli_flags_and_image_folder(tmp_path, golden_dir):
    """Flags and defaults of src/hair_retrieval.py:8-57 (paths excepted: the reference's defaults are the authors'
    private NAS locations), torchvision ImageFolder's directory contract, Resize(224)'s output size rule."""
    from PIL import Image
    spec = importlib.util.spec_from_file_location(
        "hair_cli", os.path.join(ROOT, "hair-centric-image-retrieval_amd", "hair_retrieval.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    a = cli.parse_args([])
    assert (a.model_name, a.batch_size, a.num_workers, a.device, a.embed_save_dir, a.query_image, a.top_k,
            a.num_queries, a.save_visualization, a.vis_save_dir, a.random_seed, a.extract_only, a.retrieve_only,
            a.force_extract) == ("vit_base_patch16", 64, 8, None, "save/embeddings", None, 5, 5, False,
                                 "save/visualizations", 42, False, False, False)
    with pytest.raises(SystemExit):
        cli.parse_args(["--model_name", "resnet50"])
    from hcir.hair_encoder import ImageFolder, resize_shorter_side
    win = np.load(os.path.join(golden_dir, "asset_windows.npz"))["windows"]
    for cls, names in (("b_cls", ["z.png", "a.jpg"]), ("a_cls", ["m.png", "notes.txt"])):
        os.makedirs(tmp_path / cls)
        for i, n in enumerate(names):
            if n.endswith(".txt"):
                (tmp_path / cls / n).write_text("x")
            else:
                Image.fromarray(win[i]).save(tmp_path / cls / n)
    ds = ImageFolder(str(tmp_path))
    assert ds.classes == ["a_cls", "b_cls"]
    assert [(os.path.relpath(p, tmp_path), c) for p, c in ds.samples] == [
        ("a_cls/m.png", 0), ("b_cls/a.jpg", 1), ("b_cls/z.png", 1)]
    assert resize_shorter_side(Image.new("RGB", (640, 301)), 224).size == (int(224 * 640 / 301), 224)
    assert resize_shorter_side(Image.new("RGB", (100, 333)), 224).size == (224, int(224 * 333 / 100))
    assert resize_shorter_side(Image.new("RGB", (224, 500)), 224).size == (224, 500)


def test_topk_record_pack_roundtrip():
    """One collective carries (fp32 value, int64 index) as an int32 record [Q, 3k] (hcir.dist.pack_topk)."""
    from hcir.dist import pack_topk, unpack_topk
    val = torch.tensor([[1.5, -0.0, float("-inf")], [3e-39, -2.25, float("inf")]])
    idx = torch.tensor([[0, 2 ** 40 + 7, -1], [999_999, 2 ** 31, -1]], dtype=torch.int64)
    rec = pack_topk(val, idx)
    assert rec.dtype == torch.int32 and tuple(rec.shape) == (2, 9)
    v2, i2 = unpack_topk(torch.stack([rec, rec], 0), 3)
    assert v2.shape == (2, 2, 3) and torch.equal(i2[1], idx)
    assert torch.equal(v2[0].view(torch.int32), val.view(torch.int32))     # bit-for-bit, -0.0 and denormals kept
