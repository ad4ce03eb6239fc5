"""GPU parity of the device Resize(224, bicubic) + CenterCrop(224) (src/models/hair_encoder.py:44-48) through the
C ABI (hcir_resize_bicubic_coeffs on the host, hcir_resize_crop_bicubic_u8 on the device).  Bar: BYTE-EXACT against
live Pillow `Image.resize(BICUBIC)` followed by the centre crop, and against oracle/resize.py."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _img(rng, h, w):
    base = rng.integers(0, 256, (max(h // 9, 2), max(w // 9, 2), 3)).astype(np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR)).astype(np.int16)
    a += rng.integers(-30, 30, a.shape, dtype=np.int16)
    return np.clip(a, 0, 255).astype(np.uint8)


def _ref_window(a, size=224):
    from hcir.hair_encoder import resize_shorter_side
    from hcir.transform import center_window_u8
    return center_window_u8(resize_shorter_side(Image.fromarray(a), size), size).numpy()


def test_resize_center_crop_byte_exact_sweep(hcir_built):
    from hcir import resize
    from oracle import resize as orz
    rng = np.random.default_rng(3)
    shapes = [(1024, 1024), (300, 451), (97, 61), (224, 500), (500, 224), (33, 47), (640, 480), (480, 640), (224, 224),
              (225, 223), (1200, 900), (50, 1000), (223, 1500), (2048, 1536)]
    imgs = [_img(rng, h, w) for h, w in shapes]
    dev = [torch.from_numpy(a).cuda() for a in imgs]
    out = resize.resize_center_crop(dev, 224).cpu().numpy()  # one batch of mixed sizes (up- and down-scales)
    for k, a in enumerate(imgs):
        np.testing.assert_array_equal(out[k], _ref_window(a), err_msg=f"{shapes[k]}")
    for k in (1, 2, 5):  # against the pinned restatement too
        h, w = shapes[k]
        oh, ow = orz.resize_output_size(h, w, 224)
        r = orz.resize(imgs[k], oh, ow)
        top, left = int(round((oh - 224) / 2.0)), int(round((ow - 224) / 2.0))
        np.testing.assert_array_equal(out[k], r[top:top + 224, left:left + 224])
    # a 4-D batch tensor of one size, and explicit output sizes (full Image.resize semantics, window = image)
    b4 = torch.from_numpy(np.stack([_img(rng, 400, 380) for _ in range(5)])).cuda()
    o4 = resize.resize_center_crop(b4, 224).cpu().numpy()
    for k in range(5):
        np.testing.assert_array_equal(o4[k], _ref_window(b4[k].cpu().numpy()))
    a = imgs[1]
    full = resize.resize_center_crop([dev[1]], 224, out_sizes=[(224, 224)]).cpu().numpy()[0]
    np.testing.assert_array_equal(full, np.asarray(Image.fromarray(a).resize((224, 224), Image.BICUBIC)))


def test_hair_encoder_windows_on_device(tmp_path, golden_dir, hcir_built):
    """HairEncoder.device_windows: PNG / JPEG files decoded whole on the device, a BMP on the host, then the device
    resize: the same bytes as the reference's host transform (PIL decode -> Resize(224, bicubic) -> CenterCrop)."""
    from hcir.hair_encoder import HairEncoder
    rng = np.random.default_rng(5)
    z = np.load(os.path.join(golden_dir, "png_streams.npz"))
    names = [str(n) for n in z["names"]]
    files = []
    for n in ("asset_20519_hair.png", "mixed_filters_rgba_l9", "grey_l1", "palette", "small_100x80", "far_matches"):
        i = names.index(n)
        files.append(z["data"][z["offsets"][i]:z["offsets"][i + 1]].tobytes())
    for h, w, fmt, kw in ((700, 520, "JPEG", {"quality": 90}), (300, 410, "JPEG", {"quality": 75, "progressive": True}),
                          (260, 330, "BMP", {}), (520, 700, "PNG", {})):
        b = io.BytesIO()
        Image.fromarray(_img(rng, h, w)).save(b, fmt, **kw)
        files.append(b.getvalue())
    torch.manual_seed(0)
    enc = HairEncoder(None, "vit_base_patch16", device="cuda")
    got = enc.device_windows(files).cpu().numpy()
    for k, f in enumerate(files):
        ref = enc._window_u8(Image.open(io.BytesIO(f)).convert("RGB")).numpy()
        np.testing.assert_array_equal(got[k], ref, err_msg=f"file {k}")
    # encode_single_image reads the file itself
    p = tmp_path / "q.png"
    p.write_bytes(files[0])
    e = enc.encode_single_image(str(p))
    assert e.shape == (768,) and np.isfinite(e).all()
