"""GPU parity of the device PNG decoder through the C ABI: hcir_png_stage_batch (host) ->
hcir_png_decode_window_u8 (device inflate + unfilter).  Bar: BYTE-EXACT against Pillow — the committed golden
windows (tests/golden/png_streams.npz: the reference's four hair-region PNGs + forced-filter / forced-block-type
synthetic files) and live Pillow decodes of seeded files — and against oracle/png.py, the pinned CPU restatement."""
import io
import os
import struct
import sys
import zlib

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from png_writer import chunk, filter_rows, synth_image, write_png  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def streams(golden_dir):
    z = np.load(os.path.join(golden_dir, "png_streams.npz"))
    files = [z["data"][z["offsets"][i]:z["offsets"][i + 1]].tobytes() for i in range(len(z["names"]))]
    return [str(n) for n in z["names"]], files, z["windows"]


def _window(rgb, wh, ww):
    h, w = rgb.shape[:2]
    ph, pw = max(wh - h, 0), max(ww - w, 0)
    if ph or pw:
        rgb = np.pad(rgb, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        h, w = rgb.shape[:2]
    top, left = int(round((h - wh) / 2.0)), int(round((w - ww) / 2.0))
    return rgb[top:top + wh, left:left + ww]


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def _decode(files, size=224, **kw):
    from hcir import png
    staged = png.stage_batch(files)
    out = png.decode_windows(staged.to("cuda"), size, check_status=True, **kw)
    return out.cpu().numpy(), staged


def test_golden_files_byte_exact(streams, hcir_built):
    names, files, wins = streams
    got, staged = _decode(files)  # one batch: four 1024^2 assets + 22 files of mixed size, colour type, block type
    assert staged.rejected == []
    for i, n in enumerate(names):
        np.testing.assert_array_equal(got[i], wins[i], err_msg=n)


def test_each_file_alone_and_oracle(streams, hcir_built):
    from oracle import png as op
    names, files, wins = streams
    for n, f, w in zip(names, files, wins):
        got, _ = _decode([f])
        np.testing.assert_array_equal(got[0], w, err_msg=n)
        np.testing.assert_array_equal(got[0], op.center_window(op.decode(f), 224), err_msg=n)


def test_window_sizes_and_whole_image(streams, hcir_built):
    names, files, _ = streams
    sel = [names.index(n) for n in ("asset_20521_hair.png", "mixed_filters_rgba_l9", "small_100x80", "far_matches",
                                    "tall_narrow_3x900", "wide_short_1500x5", "one_pixel", "palette")]
    for size in ((224, 224), (100, 60), (61, 333), (512, 512), (1, 1)):
        got, _ = _decode([files[i] for i in sel], size)
        for k, i in enumerate(sel):
            np.testing.assert_array_equal(got[k], _window(_pil(files[i]), *size), err_msg=f"{names[i]} {size}")
    for n in ("asset_20520_hair.png", "tall_paeth_multi_idat"):  # window = image: the whole decode
        data = files[names.index(n)]
        ref = _pil(data)
        got, _ = _decode([data], ref.shape[:2])
        np.testing.assert_array_equal(got[0], ref, err_msg=n)


def _pillow_png(rng, mode, h, w, **kw):
    c = {"RGB": 3, "RGBA": 4, "L": 1, "LA": 2, "P": 1}[mode]
    a = synth_image(rng, h, w, c, "mixed" if rng.random() < 0.8 else "noise")
    if mode == "P":
        im = Image.fromarray(a[:, :, 0], "P")
        im.putpalette(rng.integers(0, 256, 768).astype(np.uint8).tobytes())
    else:
        im = Image.fromarray(a if c > 1 else a[:, :, 0], mode)
    b = io.BytesIO()
    im.save(b, "PNG", **kw)
    return b.getvalue()


def test_live_pillow_sweep(hcir_built):
    """Files written by Pillow's own encoder (its filter heuristics, zlib levels 0-9, optimize) at random sizes."""
    rng = np.random.default_rng(11)
    files = []
    for i in range(40):
        mode = ["RGB", "RGBA", "L", "LA", "P"][i % 5]
        h, w = int(rng.integers(1, 420)), int(rng.integers(1, 420))
        kw = {"compress_level": int(rng.integers(0, 10))} if i % 3 else {"optimize": True}
        files.append(_pillow_png(rng, mode, h, w, **kw))
    got, staged = _decode(files)
    assert staged.rejected == []
    for i, f in enumerate(files):
        np.testing.assert_array_equal(got[i], _window(_pil(f), 224, 224), err_msg=f"file {i}")


@pytest.mark.parametrize("count", [256, 704])
def test_big_batches_and_determinism(streams, hcir_built, count):
    """256 files: three wavefronts per image (lookup | decoder | copier); 704 files: the full-chip form, two per image."""
    names, files, wins = streams
    order = np.random.default_rng(2).integers(0, len(files), count)
    batch = [files[i] for i in order]
    a, _ = _decode(batch)
    b, _ = _decode(batch)
    np.testing.assert_array_equal(a, b)
    for k, i in enumerate(order):
        np.testing.assert_array_equal(a[k], wins[i], err_msg=f"{k}: {names[i]}")


def _rebuild(data, new_idat):
    """the same file with another IDAT payload (CRCs valid, so only the decoder can notice)"""
    i = data.index(b"IDAT") - 4
    return data[:i] + chunk(b"IDAT", new_idat) + chunk(b"IEND", b"")


def test_corrupt_streams_are_flagged_not_fatal(streams, hcir_built):
    from hcir import png
    names, files, wins = streams
    good = files[names.index("filter4_rgb")]
    rng = np.random.default_rng(4)
    img = synth_image(rng, 261, 297, 3)
    raw = filter_rows(img, [4] * 261)
    z = zlib.compress(raw, 6)
    cases = {
        "truncated": _rebuild(good, z[:len(z) // 3]),
        "bad block type": _rebuild(good, b"\x78\x9c\x07" + bytes(40)),
        "distance before start": _rebuild(good, b"\x78\x9c\x03\x02\x00" + bytes(40)),
        "bad zlib header": _rebuild(good, b"\x79\x9c" + z[2:]),
        "filter type 7": _rebuild(good, zlib.compress(raw[:200 * 892] + b"\x07" + raw[200 * 892 + 1:], 6)),
        "garbage": _rebuild(good, b"\x78\x9c" + bytes(rng.integers(0, 256, 5000).astype(np.uint8))),
    }
    batch = [good] + list(cases.values()) + [good]
    staged = png.stage_batch(batch)
    assert staged.rejected == []
    st = torch.zeros(len(batch), dtype=torch.int32, device="cuda")
    out = torch.empty((len(batch), 224, 224, 3), dtype=torch.uint8, device="cuda")
    dev = staged.to("cuda")
    L = hcir_built
    wsb = L.hcir_png_workspace_bytes(staged._host_headers.data_ptr(), staged.b, 224, 224)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    rc = L.hcir_png_decode_window_u8(dev.blob.data_ptr(), staged._host_headers.data_ptr(), staged.b, 224, 224,
                                     out.data_ptr(), st.data_ptr(), ws.data_ptr(), ws.numel(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    st = st.cpu().tolist()
    assert st[0] == 0 and st[-1] == 0
    for k, name in enumerate(cases):
        assert st[1 + k] == -1, f"{name}: status {st[1 + k]}"
    w = wins[names.index("filter4_rgb")]
    np.testing.assert_array_equal(out[0].cpu().numpy(), w)
    np.testing.assert_array_equal(out[-1].cpu().numpy(), w)
    with pytest.raises(png.HcirError, match="corrupt"):
        png.decode_windows(dev, 224, check_status=True)


def test_rejected_files_take_the_host_path(streams, hcir_built):
    from hcir import png
    names, files, wins = streams
    rng = np.random.default_rng(6)
    b = io.BytesIO()
    Image.fromarray(rng.integers(0, 65535, (300, 260)).astype(np.uint16)).save(b, "PNG")
    sixteen = b.getvalue()
    batch = [files[0], sixteen, files[5]]
    staged = png.stage_batch(batch)
    assert staged.rejected == [1]
    with pytest.raises(png.HcirError, match="outside"):
        png.decode_windows(staged.to("cuda"))
    out = png.decode_windows(staged.to("cuda"), files=batch, host_fallback_for_rejected=True, check_status=True)
    np.testing.assert_array_equal(out[0].cpu().numpy(), wins[0])
    np.testing.assert_array_equal(out[2].cpu().numpy(), wins[5])
    np.testing.assert_array_equal(out[1].cpu().numpy(), _window(_pil(sixteen), 224, 224))


def test_png_into_knn_transform_equals_reference_transform(streams, hcir_built):
    """decode -> hcir_knn_transform_u8 == the reference's knn_transform of the PIL image (HP/utils/transform.py:10-14)."""
    from hcir import png
    from hcir.transform import knn_transform, knn_transform_u8
    names, files, _ = streams
    data = files[names.index("asset_20519_hair.png")]
    win = png.decode_windows(png.stage_batch([data]).to("cuda"))
    x = knn_transform_u8(win)[0].cpu()
    ref = knn_transform(Image.open(io.BytesIO(data)).convert("RGB"))
    assert torch.equal(x, ref)


def test_loader_routes_png_and_jpeg_to_the_device(tmp_path, streams, hcir_built):
    """EncodedDataset + collate_encoded on a mixed directory (the reference's *_hair.png lists next to *_full_face
    JPEGs): both codecs decode on the device, a 16-bit PNG rides the host path; result = the reference loader's."""
    from torch.utils.data import DataLoader
    from hcir.dataloader import CustomDataset, EncodedDataset, collate_encoded
    from hcir.transform import center_window_u8
    names, files, _ = streams
    rng = np.random.default_rng(9)
    rows = []
    for i in range(14):
        if i % 3 == 0:
            name = f"{i}_face.jpg"
            a = synth_image(rng, int(rng.integers(230, 400)), int(rng.integers(230, 400)), 3)
            Image.fromarray(a).save(tmp_path / name, quality=90)
        elif i == 7:
            name = f"{i}_deep.png"
            Image.fromarray(rng.integers(0, 65535, (240, 250)).astype(np.uint16)).save(tmp_path / name)
        else:
            name = f"{i}_hair.png"
            (tmp_path / name).write_bytes(files[(i * 5) % len(files)])
        rows.append(f"{name},{i % 3}")
    (tmp_path / "a.csv").write_text("id,class\n" + "\n".join(rows) + "\n")
    ref = DataLoader(CustomDataset(str(tmp_path / "a.csv"), str(tmp_path), center_window_u8), batch_size=5)
    for workers in (0, 2):
        dev = DataLoader(EncodedDataset(str(tmp_path / "a.csv"), str(tmp_path)), batch_size=5, num_workers=workers,
                         collate_fn=collate_encoded)
        for (rw, rl), (eb, el) in zip(ref, dev):
            assert torch.equal(rl, el)
            assert any(k == "png" for k, _, _ in eb.parts)
            assert torch.equal(eb.decode("cuda").cpu(), rw)


def test_mutated_streams_never_fault_and_agree_with_pillow(streams, hcir_built):
    """384 files whose zlib streams carry random byte / bit damage (chunk CRCs re-made, so only the decoder can notice):
    the launch must come back (every loop of the kernels consumes input or leaves; the two wavefronts' polls are
    bounded), and whenever Pillow decodes a damaged file the device must report it sound and give the same window —
    damage behind the last scanline the window needs is invisible to both.  Where the device flags a file, Pillow must
    have rejected it too."""
    import warnings
    from hcir import png
    names, files, wins = streams
    rng = np.random.default_rng(77)
    bases = [files[names.index(n)] for n in ("filter4_rgb", "mixed_blocks", "grey_l1", "palette", "mixed_filters_rgba_l9",
                                             "fixed_blocks")]
    muts = []
    for k in range(384):
        f = bases[k % len(bases)]
        i = f.index(b"IDAT")
        n = struct.unpack(">I", f[i - 4:i])[0]
        z = bytearray(f[i + 4:i + 4 + n])
        kind = k % 4
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(2, len(z)))
            if kind == 0:
                z[at] ^= 1 << int(rng.integers(0, 8))          # one bit
            elif kind == 1:
                z[at] = int(rng.integers(0, 256))               # one byte
            elif kind == 2:
                z[at:at + 8] = bytes(rng.integers(0, 256, 8).astype(np.uint8))[:len(z) - at]   # a burst
            else:
                del z[at:at + int(rng.integers(1, 40))]         # bytes lost: everything behind shifts
        muts.append(f[:i - 4] + chunk(b"IDAT", bytes(z)) + chunk(b"IEND", b""))
    staged = png.stage_batch(muts)
    assert staged.rejected == []
    dev = staged.to("cuda")
    st = torch.zeros(len(muts), dtype=torch.int32, device="cuda")
    out = torch.empty((len(muts), 224, 224, 3), dtype=torch.uint8, device="cuda")
    L = hcir_built
    ws = torch.empty(L.hcir_png_workspace_bytes(staged._host_headers.data_ptr(), staged.b, 224, 224), dtype=torch.uint8,
                     device="cuda")
    assert L.hcir_png_decode_window_u8(dev.blob.data_ptr(), staged._host_headers.data_ptr(), staged.b, 224, 224,
                                       out.data_ptr(), st.data_ptr(), ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    out = out.cpu().numpy()
    sound = flagged = 0
    for k, f in enumerate(muts):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = _window(_pil(f), 224, 224)
        except Exception:  # noqa: BLE001 - Pillow rejects the file (zlib error, truncated data, bad Adler-32 ...)
            ref = None
        if ref is not None:
            sound += 1
            assert st[k] == 0, f"mutant {k}: Pillow decodes it, the device flags it"
            np.testing.assert_array_equal(out[k], ref, err_msg=f"mutant {k}")
        if st[k] != 0:
            flagged += 1
            assert ref is None
    assert flagged > 100 and st.min() >= -1
    print(f"{flagged} of {len(muts)} mutants flagged by the device, {sound} decoded by Pillow")
