"""CPU tests of the resize path (hair_retrieval's Resize(224, bicubic)): the oracle restatement (oracle/resize.py)
pinned byte for byte to live Pillow, and the library's HOST coefficient tables against the oracle's."""
import numpy as np
import pytest
from PIL import Image

SIZES = [((1024, 1024), (224, 224)), ((300, 451), (224, 336)), ((97, 61), (356, 224)), ((224, 500), (224, 500)),
         ((500, 224), (500, 224)), ((33, 47), (224, 319)), ((640, 480), (298, 224)), ((17, 1200), (224, 15811 // 70))]


def _img(rng, h, w):
    base = rng.integers(0, 256, (max(h // 9, 2), max(w // 9, 2), 3)).astype(np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR)).astype(np.int16)
    a += rng.integers(-30, 30, a.shape, dtype=np.int16)
    return np.clip(a, 0, 255).astype(np.uint8)


def test_oracle_resize_equals_pillow():
    from oracle import resize as orz
    rng = np.random.default_rng(1)
    for (h, w), (oh, ow) in SIZES:
        a = _img(rng, h, w)
        ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))
        np.testing.assert_array_equal(orz.resize(a, oh, ow), ref, err_msg=f"{h}x{w} -> {oh}x{ow}")
    for h, w in ((1024, 1024), (300, 451), (97, 61), (224, 500), (640, 480)):
        oh, ow = orz.resize_output_size(h, w, 224)
        assert min(oh, ow) == 224 and (oh, ow) == ((224, int(224 * w / h)) if h <= w else (int(224 * h / w), 224))


def test_library_tables_equal_oracle(hcir_built):
    from hcir import resize as hrz
    from oracle import resize as orz
    for i, o in ((1024, 224), (451, 336), (61, 224), (97, 356), (47, 319), (1200, 225), (224, 223), (5, 4000)):
        ks, tab = hrz.axis_table(i, o)
        oks, ob, okk = orz.coeffs(i, o)
        assert ks == oks
        np.testing.assert_array_equal(tab[:2 * o].reshape(o, 2), ob)
        np.testing.assert_array_equal(tab[2 * o:].reshape(o, ks), okk)
    assert hcir_built.hcir_resize_bicubic_ksize(0, 5) == 0
    with pytest.raises(hrz.HcirError):
        hrz.resize_center_crop([__import__("torch").zeros((4, 4, 3), dtype=__import__("torch").uint8)])  # not on a HIP device
