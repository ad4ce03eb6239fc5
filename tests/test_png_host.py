"""CPU tests of the PNG path: the oracle restatement (oracle/png_oracle.c: inflate + unfilter) pinned
byte for byte to zlib and Pillow on the golden files, and the library's HOST entry points (chunk walk, CRC-32,
staging copy — no GPU call)."""
import ctypes
import io
import os
import struct
import sys
import zlib

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from png_writer import chunk, synth_image, write_png  # noqa: E402


@pytest.fixture(scope="module")
def streams(golden_dir):
    z = np.load(os.path.join(golden_dir, "png_streams.npz"))
    files = [z["data"][z["offsets"][i]:z["offsets"][i + 1]].tobytes() for i in range(len(z["names"]))]
    return [str(n) for n in z["names"]], files, z["windows"], z["asset_strips"]


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_oracle_inflate_equals_zlib(streams):
    from oracle import png as op
    names, files, _, _ = streams
    for name, data in zip(names, files):
        idat = op.parse(data)["idat"]
        want = zlib.decompress(idat)
        assert op.inflate(idat, len(want) + 10) == want, name
        cut = len(want) * 2 // 3
        assert op.inflate(idat, cut) == want[:cut], name  # what a window decode asks for: stop mid-stream
    rng = np.random.default_rng(3)
    for level in (0, 1, 6, 9):
        for n in (0, 1, 5, 70000):
            raw = bytes(rng.integers(0, 7, n).astype(np.uint8))
            assert op.inflate(zlib.compress(raw, level), n + 1) == raw


def test_oracle_decode_equals_pillow_and_goldens(streams):
    from oracle import png as op
    names, files, wins, strips = streams
    k = 0
    for name, data, win in zip(names, files, wins):
        rgb = op.decode(data)
        np.testing.assert_array_equal(rgb, _pil(data), err_msg=name)  # live Pillow
        np.testing.assert_array_equal(op.center_window(rgb, 224), win, err_msg=name)  # committed golden
        if name.startswith("asset_"):
            np.testing.assert_array_equal(rgb[:320:5, ::3], strips[k], err_msg=name)
            k += 1


def test_oracle_rejects_corrupt_streams(streams):
    from oracle import png as op
    names, files, _, _ = streams
    idat = op.parse(files[names.index("filter4_rgb")])["idat"]
    with pytest.raises(op.Corrupt):
        op.inflate(idat[:len(idat) // 2], 10 ** 7)  # ends early
    bad = bytearray(idat)
    bad[-1] ^= 1  # Adler-32
    with pytest.raises(op.Corrupt):
        op.inflate(bytes(bad), 10 ** 7)
    with pytest.raises(op.Corrupt):
        op.inflate(b"\x78\x9c\x07", 10)  # block type 3
    # a match that reaches before the start of the data: fixed block, length 3, distance 1 as the first symbol
    with pytest.raises(op.Corrupt):
        op.inflate(b"\x78\x9c\x03\x02\x00", 10)


def _hdr_of(L, data, flags=1):
    from hcir import png
    hdr = png.PngHeader()
    a = np.frombuffer(data, np.uint8)
    need = L.hcir_png_stage_bytes(a.ctypes.data, a.size)
    blob = np.full(need + 64, 0xAB, np.uint8)
    used = ctypes.c_size_t(0)
    rc = L.hcir_png_stage(a.ctypes.data, a.size, flags, ctypes.byref(hdr), blob.ctypes.data, 16, blob.size,
                          ctypes.byref(used))
    return rc, hdr, blob, used.value, need


def test_stage_host(streams, hcir_built):
    """hcir_png_stage: header fields, the staged bytes = the IDAT payloads joined, zero padded."""
    from oracle import png as op
    names, files, _, _ = streams
    L = hcir_built
    for name, data in zip(names, files):
        rc, hdr, blob, used, need = _hdr_of(L, data)
        assert rc == 0, name
        p = op.parse(data)
        assert (hdr.width, hdr.height, hdr.color_type) == (p["width"], p["height"], p["color_type"]), name
        assert hdr.bpp == op.bytes_per_pixel(p["color_type"]) and hdr.stage_offset == 16
        assert hdr.stream_bytes == len(p["idat"]) and used == need and used % 16 == 0 and used >= len(p["idat"]) + 16
        assert blob[16:16 + len(p["idat"])].tobytes() == p["idat"], name
        assert not blob[16 + len(p["idat"]):16 + used].any(), "padding behind the stream must be zero"
        if p["color_type"] == 3:
            n = np.flatnonzero(np.frombuffer(bytes(hdr.palette), np.uint8)).max() + 1
            assert bytes(hdr.palette)[:n] == p["palette"][:n].tobytes()


def test_stage_rejects(hcir_built):
    L = hcir_built
    rng = np.random.default_rng(5)
    img = synth_image(rng, 40, 50, 3)
    good = write_png(img, 2, 4)
    assert _hdr_of(L, good)[0] == 0
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x40  # inside IDAT: the chunk's CRC no longer matches
    assert _hdr_of(L, bytes(bad))[0] == -1  # HCIR_ERR_INVALID, as Pillow raises on a bad checksum
    assert _hdr_of(L, bytes(bad), flags=0)[0] == 0  # the caller may skip the CRCs
    assert _hdr_of(L, b"\xff\xd8\xff\xe0" + bytes(60))[0] == -1  # not a PNG
    assert _hdr_of(L, good[:40])[0] == -1  # truncated
    # valid PNGs outside the device subset: HCIR_ERR_UNSUPPORTED (-2), the loader keeps PIL for them
    b = io.BytesIO()
    Image.fromarray((rng.integers(0, 65535, (20, 20))).astype(np.uint16)).save(b, "PNG")
    assert _hdr_of(L, b.getvalue())[0] == -2  # 16-bit
    b = io.BytesIO()
    Image.fromarray(rng.integers(0, 2, (20, 20)).astype(np.uint8) * 255).convert("1").save(b, "PNG")
    assert _hdr_of(L, b.getvalue())[0] == -2  # 1-bit
    ihdr = struct.pack(">IIBBBBB", 50, 40, 8, 2, 0, 0, 1)  # Adam7
    inter = good[:8] + chunk(b"IHDR", ihdr) + good[8 + 25:]
    assert _hdr_of(L, inter)[0] == -2
    wide = write_png(synth_image(rng, 2, 8200, 1), 0, 0)
    assert _hdr_of(L, wide)[0] == -2  # wider than the device's line buffer


def test_stage_batch_and_workspace(streams, hcir_built):
    from hcir import png
    names, files, _, _ = streams
    jpeg_like = b"\xff\xd8" + bytes(100)
    st = png.stage_batch(files + [jpeg_like], pin=False, threads=4)
    assert st.rejected == [len(files)] and st.b == len(files) + 1
    hs = st.headers()
    assert hs[len(files)].width == 0
    for h, data in zip(hs, files):
        p = Image.open(io.BytesIO(data))
        assert (h.width, h.height) == p.size
        assert st.blob[h.stage_offset:h.stage_offset + 2].tolist() == [0x78, 0x9c] or h.stream_bytes > 0
    L = hcir_built
    ws = L.hcir_png_workspace_bytes(st._host_headers.data_ptr(), st.b, 224, 224)
    # every image's scanlines 0..last needed row fit: the 1024^2 assets need 624 rows of 3073 bytes
    assert ws >= st.b * 624 * 3073
    assert L.hcir_png_workspace_bytes(st._host_headers.data_ptr(), st.b, 0, 224) == 0
    with pytest.raises(png.HcirError):
        png.decode_windows(st)  # a host blob: the decoder is device-only, no CPU fallback


def test_stage_batch_recycled_blob_and_file_sniffing(streams, hcir_built):
    """A loader recycles its staging blobs (stage_batch(out=...)): same bytes as a fresh blob, a blob that is too small
    is replaced; HairEncoder groups its files by a sniff of the first bytes (codec, height, width) without a decoder."""
    import torch
    from hcir import png
    from hcir.hair_encoder import _sniff
    names, files, _, _ = streams
    fresh = png.stage_batch(files, pin=False, threads=4)
    big = torch.full((fresh.blob.numel() + 4096,), 0x5A, dtype=torch.uint8)
    again = png.stage_batch(files, pin=False, threads=2, out=big)
    assert again.blob.data_ptr() == big.data_ptr() and again.blob.numel() == fresh.blob.numel()
    np.testing.assert_array_equal(again.status, fresh.status)
    for hf, ha in zip(fresh.headers(), again.headers()):
        assert bytes(hf) == bytes(ha)
        if hf.width:
            np.testing.assert_array_equal(again.blob.numpy()[ha.stage_offset:ha.stage_offset + ha.stream_bytes],
                                          fresh.blob.numpy()[hf.stage_offset:hf.stage_offset + hf.stream_bytes])
    repl = png.stage_batch(files, pin=False, out=torch.zeros(64, dtype=torch.uint8))
    assert repl.blob.numel() == fresh.blob.numel()
    for data, h in zip(files, fresh.headers()):
        kind, hh, ww = _sniff(np.frombuffer(data, np.uint8))
        with Image.open(io.BytesIO(data)) as im:
            assert (kind, ww, hh) == ("png", im.size[0], im.size[1])
    buf = io.BytesIO()
    Image.fromarray(np.zeros((33, 47, 3), np.uint8)).save(buf, "JPEG", quality=70, progressive=True)
    assert _sniff(np.frombuffer(buf.getvalue(), np.uint8)) == ("jpeg", 33, 47)
    buf = io.BytesIO()
    Image.fromarray(np.zeros((8, 9, 3), np.uint8)).save(buf, "BMP")
    assert _sniff(np.frombuffer(buf.getvalue(), np.uint8))[0] == "host"
    assert _sniff(np.frombuffer(b"\x89PNG\r\n\x1a\n" + bytes(4), np.uint8))[0] == "host"   # cut short
    assert _sniff(np.frombuffer(b"\xff\xd8\xff\xe0", np.uint8))[0] == "host"


def test_stager_fuzz_under_sanitizers(streams):
    """The host stager (png_stage.h) against 20 000 mutated / truncated files under AddressSanitizer + UBSan."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "libpng_stage_fuzz.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", os.path.join(root, "tests", "png_stage_fuzz.cpp"), "-o", exe])
    names, files, _, _ = streams
    driver = (
        "import ctypes, sys, numpy as np\n"
        "L = ctypes.CDLL(sys.argv[1])\n"
        "z = np.load(sys.argv[2])\n"
        "tot = 0\n"
        "for i in (4, 9, 12, 13, 16, 20, 23):\n"
        "    f = z['data'][z['offsets'][i]:z['offsets'][i + 1]].copy()\n"
        "    acc = ctypes.c_int64(0)\n"
        "    rc = L.png_stage_fuzz(f.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(f.size), 3000, ctypes.byref(acc))\n"
        "    assert rc == 0, rc\n"
        "    tot += acc.value\n"
        "print('accepted', tot)\n")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", driver, exe, os.path.join(root, "tests", "golden", "png_streams.npz")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "accepted" in r.stdout
