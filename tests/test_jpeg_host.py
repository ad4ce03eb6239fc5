"""CPU tests of the JPEG path: the oracle restatement against Pillow's libjpeg-turbo and the golden
streams, the library's HOST entry points (marker walk, staging copy — no GPU call), and the device code's
arithmetic compiled for the host (tests/jpeg_emul.cpp: the kernels' phases as loops over emulated threads)."""
import ctypes
import io
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def streams(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_streams.npz"))
    files = [z["data"][z["offsets"][i]:z["offsets"][i + 1]].tobytes() for i in range(len(z["names"]))]
    return [str(n) for n in z["names"]], files, z["windows"]


@pytest.fixture(scope="module")
def emul():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libjpeg_emul.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas",
                           os.path.join(ROOT, "tests", "jpeg_emul.cpp"), "-o", so])
    L = ctypes.CDLL(so)
    L.emul_header_bytes.restype = ctypes.c_size_t
    L.emul_decode_window.restype = ctypes.c_int
    L.emul_decode_window.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_void_p, ctypes.c_void_p]

    def decode(data, wh, ww, threads=1024):
        out = np.zeros((wh, ww, 3), np.uint8)
        st = np.zeros(4, np.int32)
        rc = L.emul_decode_window(data, len(data), wh, ww, threads, out.ctypes.data, st.ctypes.data)
        return rc, out, st

    decode.lib = L
    return decode


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


def _synth(rng, h, w):
    base = rng.integers(0, 256, (h // 16 + 2, w // 16 + 2, 3)).astype(np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.int16)
    a += rng.integers(-20, 20, a.shape, dtype=np.int16)
    return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))


def _encode(im, **kw):
    b = io.BytesIO()
    im.save(b, "JPEG", **kw)
    return b.getvalue()


def test_oracle_matches_golden_windows(streams):
    """oracle/jpeg.py (bit-serial Huffman, numpy islow IDCT, fancy upsampling, jdcolor tables) reproduces the
    committed Pillow / libjpeg-turbo 3.1.4 windows byte for byte — the pin of the JPEG oracle."""
    from oracle import jpeg as oj
    names, files, wins = streams
    n = 0
    for name, data, win in zip(names, files, wins):
        if name.startswith("reject/"):
            with pytest.raises(oj.Unsupported):
                oj.decode(data)
            continue
        if name.startswith("asset/") and n >= 2 and not name.endswith("20522.jpg"):
            continue  # pure-Python Huffman: ~2 s per 1024^2 stream; three of the four keep the suite short
        np.testing.assert_array_equal(oj.center_window(oj.decode(data)), win, err_msg=name)
        n += 1
    assert n >= 14


def test_oracle_matches_live_pillow():
    from oracle import jpeg as oj
    rng = np.random.default_rng(5)
    for (h, w) in [(64, 64), (50, 70), (33, 17), (8, 8), (1, 1), (17, 3), (9, 2), (3, 5)]:
        for ss in (0, 1, 2):
            for ri in (0, 3):
                kw = dict(quality=int(rng.integers(30, 100)), subsampling=ss)
                if ri:
                    kw["restart_marker_blocks"] = ri
                data = _encode(_synth(rng, h, w), **kw)
                np.testing.assert_array_equal(oj.decode(data), _pil(data), err_msg=f"{h}x{w} ss{ss} ri{ri}")
    data = _encode(_synth(rng, 40, 56).convert("L"), quality=80)
    np.testing.assert_array_equal(oj.decode(data), _pil(data))


def test_header_struct_matches_ctypes_mirror(emul, hcir_built):
    from hcir import jpeg
    assert emul.lib.emul_header_bytes() == jpeg.HEADER_BYTES


def test_stage_host_entry_points(streams, hcir_built):
    """hcir_jpeg_stage / _stage_batch are host functions: geometry, tables and the unstuffed stream against the
    oracle's parse; rejected files carry a status and a zeroed header."""
    from hcir import jpeg
    from oracle import jpeg as oj
    names, files, _ = streams
    staged = jpeg.stage_batch(files, pin=False, threads=4)
    hdrs = staged.headers()
    blob = staged.blob.numpy()
    for i, (name, data) in enumerate(zip(names, files)):
        if name.startswith("reject/"):
            assert staged.status[i] in (-1, -2) and hdrs[i].width == 0, name
            continue
        assert staged.status[i] == 0, name
        info = oj.parse(data)
        h = hdrs[i]
        assert (h.height, h.width, h.ncomp) == (info["height"], info["width"], info["ncomp"])
        assert h.restart_interval == info["restart_interval"]
        ref = b"".join(info["segments"])
        assert h.stream_bits == 8 * len(ref), name
        words = blob[h.stage_offset:h.stage_offset + 4 * h.stream_words].view("<u4")
        got = words.astype(">u4").tobytes()
        assert got[:len(ref)] == ref and set(got[len(ref):]) <= {0xFF}, name   # bit 31 first; 1-padding behind
        assert h.nsegments == len(info["segments"]) or info["restart_interval"] == 0
        seg = blob[h.stage_offset + (4 * h.stream_words + 15) // 16 * 16:][:4 * (h.nsegments + 1)].view("<u4")
        starts = np.cumsum([0] + [8 * len(s) for s in info["segments"]])
        np.testing.assert_array_equal(seg[:len(starts)], starts)
        for c, comp in enumerate(info["comps"] if info["ncomp"] == 3 else info["comps"][:1]):
            np.testing.assert_array_equal(np.array(h.quant[c]), info["qt"][comp["tq"]])
    assert staged.rejected == [i for i, n in enumerate(names) if n.startswith("reject/")]
    with pytest.raises(jpeg.HcirError):
        jpeg.decode_windows(staged)          # blob is on the CPU: no fallback


def test_stage_rejects_truncated_and_garbage(hcir_built):
    from hcir import jpeg
    rng = np.random.default_rng(0)
    good = _encode(_synth(rng, 64, 64), quality=90)
    staged = jpeg.stage_batch([good, good[:100], b"\xff\xd8\xff", bytes(rng.integers(0, 256, 500, dtype=np.uint8)),
                               _encode(_synth(rng, 32, 32).convert("CMYK"), quality=90)], pin=False)
    assert staged.status[0] == 0 and all(s != 0 for s in staged.status[1:]), staged.status


def test_stage_batch_recycled_blob_and_scan_end(streams, hcir_built):
    """A loader recycles its blobs: stage_batch(out=...) stages in place, byte-identical to a fresh blob; a blob that
    is too small is replaced.  The entropy-coded segment ends at the first real marker - bytes behind EOI (a second
    image, padding that looks like stuffing) must not reach the stream."""
    import torch
    from hcir import jpeg
    names, files, _ = streams
    fresh = jpeg.stage_batch(files, pin=False, threads=4)
    big = torch.full((fresh.blob.numel() + 8192,), 0xA5, dtype=torch.uint8)
    again = jpeg.stage_batch(files, pin=False, threads=3, out=big)
    assert again.blob.data_ptr() == big.data_ptr()
    np.testing.assert_array_equal(again.status, fresh.status)
    # identical wherever something is defined (the gaps between a file's bound and what it used are not written)
    hf, ha = fresh.headers(), again.headers()
    bf, ba = fresh.blob.numpy(), again.blob.numpy()
    for i in range(len(files)):
        assert bytes(hf[i]) == bytes(ha[i])
        if fresh.status[i] == 0:
            n = 4 * hf[i].stream_words                      # the words, then (16-byte aligned) the segment table
            t, tn = (n + 15) // 16 * 16, 4 * (hf[i].nsegments + 1)
            oa, of = ha[i].stage_offset, hf[i].stage_offset
            np.testing.assert_array_equal(ba[oa:oa + n], bf[of:of + n])
            np.testing.assert_array_equal(ba[oa + t:oa + t + tn], bf[of + t:of + t + tn])
    small = torch.zeros(1000, dtype=torch.uint8)
    repl = jpeg.stage_batch(files, pin=False, out=small)
    assert repl.blob.data_ptr() != small.data_ptr() and repl.blob.numel() == fresh.blob.numel()
    rng = np.random.default_rng(5)
    good = _encode(_synth(rng, 96, 80), quality=85)
    tail = good + b"\xff\x00\xff\xd0" + bytes(rng.integers(0, 256, 300, dtype=np.uint8)) + good
    a, b2 = jpeg.stage_batch([good], pin=False), jpeg.stage_batch([tail], pin=False)
    ha, hb = a.headers()[0], b2.headers()[0]
    assert hb.stream_bits == ha.stream_bits and hb.stream_words == ha.stream_words
    np.testing.assert_array_equal(b2.blob.numpy()[hb.stage_offset:hb.stage_offset + 4 * hb.stream_words],
                                  a.blob.numpy()[ha.stage_offset:ha.stage_offset + 4 * ha.stream_words])


def test_emulated_device_path_matches_golden(streams, emul):
    names, files, wins = streams
    for name, data, win in zip(names, files, wins):
        if name.startswith("reject/"):
            rc, _, _ = emul(data, 224, 224)
            assert rc in (-1, -2), name
            continue
        for threads in (1024, 96, -512):  # negative: the generic decode loop (tables outside the fast lookup form)
            rc, out, st = emul(data, 224, 224, threads)
            assert rc == 0, name
            np.testing.assert_array_equal(out, win, err_msg=f"{name} T={threads}")


def test_emulated_device_path_matches_live_pillow(emul):
    """Sizes around the MCU grid, every sampling, restart intervals down to one MCU, windows larger / smaller / equal
    to the image, 7 to 1024 emulated threads (subsequences from 128 bits up)."""
    rng = np.random.default_rng(11)
    n = 0
    for (h, w) in [(64, 64), (50, 70), (33, 17), (8, 8), (1, 1), (17, 3), (256, 240), (3, 5), (225, 223)]:
        for ss in (0, 1, 2):
            for ri in (0, 1, 5):
                for grey in (False, True):
                    if grey and ss:
                        continue
                    im = _synth(rng, h, w)
                    kw = dict(quality=int(rng.choice([30, 75, 95, 100])), optimize=bool(rng.integers(0, 2)))
                    if grey:
                        im = im.convert("L")
                    else:
                        kw["subsampling"] = ss
                    if ri:
                        kw["restart_marker_blocks"] = ri
                        kw.pop("optimize", None)  # Pillow cannot combine optimised tables with restart markers
                    data = _encode(im, **kw)
                    full = _pil(data)
                    for (wh, ww) in [(224, 224), (h, w), (32, 48)]:
                        for threads in (7, 1024, -64):
                            rc, out, _ = emul(data, wh, ww, threads)
                            assert rc == 0
                            np.testing.assert_array_equal(out, _window(full, wh, ww),
                                                          err_msg=f"{h}x{w} ss{ss} ri{ri} grey{grey} win{wh}x{ww} T{threads}")
                            n += 1
    assert n > 500


def test_stage_leaves_rgb_jpegs_to_the_host(hcir_built):
    """libjpeg's colour-space rule (jdapimin.c default_decompress_parms): a three-component file without a JFIF marker
    whose Adobe marker says transform 0 (or whose component ids are 'R','G','B') is RGB and is NOT converted.  The
    device path always converts, so the stager hands such files to the host decoder; and a DC Huffman
    table with a category above 15 is a bad table, as in jdhuff.c."""
    from hcir import jpeg
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (64, 64, 3)).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=90, keep_rgb=True)   # Adobe APP14, transform 0, no JFIF
    assert b"Adobe" in b.getvalue() and b"JFIF" not in b.getvalue()
    assert jpeg.stage_batch([b.getvalue()], pin=False).status.tolist() == [-2]
    b2 = io.BytesIO()
    Image.fromarray(a).save(b2, "JPEG", quality=90)
    good = b2.getvalue()
    assert jpeg.stage_batch([good], pin=False).status.tolist() == [0]
    # corrupt the first DC table's first symbol to 16
    i = good.index(b"\xff\xc4")
    assert good[i + 4] >> 4 == 0  # a DC table
    bad = bytearray(good)
    bad[i + 5 + 16] = 16
    assert jpeg.stage_batch([bytes(bad)], pin=False).status.tolist() == [-1]
