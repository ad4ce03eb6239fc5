"""hcir.jpeg — baseline-JPEG decode of CenterCrop windows on the HIP device (include/hcir.h, csrc/jpeg.hip).

Stands where the reference decodes on the host in front of `knn_transform`:
`read_file` + `torchvision.io.decode_image(img_bytes, mode=RGB)` (HP/utils/dataloader.py:28-31) and
`Image.open(path).convert('RGB')` (src/models/hair_encoder.py:108,169).  Output bytes equal libjpeg-turbo's
default decompressor (what both link), so the rest of the path (`knn_transform_u8` -> ViT) sees the same pixels.

    staged = stage_batch([bytes, ...])            # host: marker walk + unstuffing copy into ONE pinned blob
    win    = decode_windows(staged.to(device))    # device: [B, 224, 224, 3] uint8, the CenterCrop(224) windows

Files outside the baseline subset (progressive, CMYK, PNG ...) are reported in `staged.rejected`; `decode_windows`
fills their windows from `host_window(...)` (PIL, as the reference does) when `host_fallback_for_rejected=True`,
else raises.  There is no CPU path for the supported files: `decode_windows` refuses a non-HIP blob.
"""
from __future__ import annotations

import ctypes
import io
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import HcirError, check


class _Lut(ctypes.Structure):
    _fields_ = [("look", ctypes.c_uint16 * 2048), ("look2", ctypes.c_uint16 * 256), ("base2", ctypes.c_uint32),
                ("use2", ctypes.c_uint32)]


class _HuffTab(ctypes.Structure):
    _fields_ = [("lut", _Lut), ("limit", ctypes.c_uint32 * 18), ("valoff", ctypes.c_int32 * 17),
                ("vals", ctypes.c_uint8 * 256), ("is_ac", ctypes.c_uint32)]


class JpegHeader(ctypes.Structure):
    """Mirror of hcir_jpeg_header (include/hcir.h)."""
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("hmax", ctypes.c_int32), ("vmax", ctypes.c_int32), ("hs", ctypes.c_int32 * 3),
                ("vs", ctypes.c_int32 * 3), ("mcus_x", ctypes.c_int32), ("mcus_y", ctypes.c_int32),
                ("blocks_per_mcu", ctypes.c_int32), ("restart_interval", ctypes.c_int32),
                ("nsegments", ctypes.c_int32), ("stream_bits", ctypes.c_uint32), ("stream_words", ctypes.c_uint32),
                ("stage_offset", ctypes.c_uint64), ("blk_comp", ctypes.c_uint8 * 12), ("dc_tab", ctypes.c_uint8 * 4),
                ("ac_tab", ctypes.c_uint8 * 4), ("quant", (ctypes.c_uint16 * 64) * 3), ("huff", _HuffTab * 4)]


HEADER_BYTES = ctypes.sizeof(JpegHeader)
Bytes = Union[bytes, bytearray, memoryview, np.ndarray, torch.Tensor]


def _as_u8(buf: Bytes) -> np.ndarray:
    if isinstance(buf, torch.Tensor):
        buf = buf.numpy()
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.dtype != np.uint8 or a.ndim != 1:
        raise HcirError("a JPEG file is a 1-D uint8 buffer")
    return np.ascontiguousarray(a)


class StagedBatch:
    """One staging blob: `b` headers, then every image's unstuffed entropy stream + restart-segment table."""

    def __init__(self, blob: torch.Tensor, b: int, status: np.ndarray, host_headers: Optional[torch.Tensor] = None):
        self.blob, self.b, self.status = blob, b, status
        # the launcher sizes its grids from the headers on the HOST; keep a host copy once the blob is on the device
        self._host_headers = host_headers if host_headers is not None else blob[:b * HEADER_BYTES]

    @property
    def rejected(self) -> List[int]:
        return [int(i) for i in np.nonzero(self.status != 0)[0]]

    def headers(self):
        return (JpegHeader * self.b).from_address(self._host_headers.data_ptr())

    def sizes(self) -> List[Tuple[int, int]]:
        return [(h.height, h.width) for h in self.headers()]

    def stream_bytes(self) -> int:
        return int(sum(h.stream_bits for h in self.headers()) // 8)

    def to(self, device, non_blocking: bool = True) -> "StagedBatch":
        dev = torch.device(device)
        if self.blob.device == dev:
            return self
        return StagedBatch(self.blob.to(dev, non_blocking=non_blocking), self.b, self.status, self._host_headers)

    def pin(self) -> "StagedBatch":
        if self.blob.is_cuda or self.blob.is_pinned():
            return self
        return StagedBatch(self.blob.pin_memory(), self.b, self.status)


def stage_batch(files: Sequence[Bytes], pin: Optional[bool] = None, threads: int = 8,
                out: Optional[torch.Tensor] = None) -> StagedBatch:
    """Parse + stage every file into one blob (hcir_jpeg_stage_batch).  Host only; needs no GPU.
    `out`: a host uint8 blob to stage into (a loader recycles its pinned blobs: no allocation, one pass over the
    files); a blob that is too small is replaced by a fresh one."""
    L = _lib.lib()
    arrs = [_as_u8(f) for f in files]
    b = len(arrs)
    if b == 0:
        raise HcirError("stage_batch needs at least one file")
    ptrs = (ctypes.c_void_p * b)(*[a.ctypes.data for a in arrs])
    lens = (ctypes.c_size_t * b)(*[a.size for a in arrs])
    status = np.zeros(b, dtype=np.int32)
    used = ctypes.c_size_t(0)
    if out is not None:
        if out.is_cuda or out.dtype != torch.uint8 or not out.is_contiguous():
            raise HcirError("stage_batch(out=...) needs a contiguous host uint8 tensor")
        rc = L.hcir_jpeg_stage_batch(ptrs, lens, b, out.data_ptr(), out.numel(), ctypes.byref(used),
                                     status.ctypes.data, threads)
        if rc == 0:
            return StagedBatch(out[:used.value], b, status)
        if rc != -4:                      # anything but HCIR_ERR_WORKSPACE
            check(rc, "hcir_jpeg_stage_batch")
    else:
        check(L.hcir_jpeg_stage_batch(ptrs, lens, b, None, 0, ctypes.byref(used), status.ctypes.data, threads),
              "hcir_jpeg_stage_batch(size)")
    if pin is None:
        pin = torch.cuda.is_available()
    blob = torch.empty(used.value, dtype=torch.uint8, pin_memory=bool(pin))
    check(L.hcir_jpeg_stage_batch(ptrs, lens, b, blob.data_ptr(), blob.numel(), ctypes.byref(used),
                                  status.ctypes.data, threads), "hcir_jpeg_stage_batch")
    return StagedBatch(blob, b, status)


def host_window(file: Bytes, size: Tuple[int, int] = (224, 224)) -> torch.Tensor:
    """The reference's own host decode (PIL) + CenterCrop window, for files the device path does not take."""
    from PIL import Image
    from .transform import center_window_u8
    if size[0] != size[1]:
        raise HcirError("host_window supports square windows")
    with Image.open(io.BytesIO(_as_u8(file).tobytes())) as im:
        return center_window_u8(im, size[0])


_ws = {}


def decode_windows(staged: StagedBatch, size: Union[int, Tuple[int, int]] = 224, files: Optional[Sequence[Bytes]] = None,
                   host_fallback_for_rejected: bool = False, check_status: bool = False,
                   _skip_rejected_check: bool = False) -> torch.Tensor:
    """[B, win_h, win_w, 3] uint8 on the blob's device: CenterCrop(size) of every decoded image (zero where the
    window leaves the image, as torchvision pads).  Asynchronous on the current stream unless `check_status`."""
    if not staged.blob.is_cuda:
        raise HcirError(f"the staging blob is on {staged.blob.device}; hcir_jpeg_decode_window_u8 runs on a HIP "
                        "device only (no CPU fallback) — call staged.to(device) first")
    win_h, win_w = (size, size) if isinstance(size, int) else size
    L = _lib.lib()
    dev = staged.blob.device
    hdrs = staged._host_headers.data_ptr()
    out = torch.empty((staged.b, win_h, win_w, 3), dtype=torch.uint8, device=dev)
    wsb = L.hcir_jpeg_workspace_bytes(hdrs, staged.b, win_h, win_w)
    if wsb == 0:
        raise HcirError("hcir_jpeg_workspace_bytes: invalid headers")
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < wsb:
        ws = _ws[key] = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.empty(staged.b, dtype=torch.int32, device=dev) if check_status else None
    check(L.hcir_jpeg_decode_window_u8(staged.blob.data_ptr(), hdrs, staged.b, win_h, win_w, out.data_ptr(),
                                       None if st is None else st.data_ptr(), ws.data_ptr(), ws.numel(), stream),
          "hcir_jpeg_decode_window_u8")
    rej = [] if _skip_rejected_check else staged.rejected  # a caller that fills those windows itself
    if rej:
        if not host_fallback_for_rejected or files is None:
            raise HcirError(f"files {rej} are outside the device decoder's baseline-JPEG subset "
                            "(pass files= and host_fallback_for_rejected=True to decode those on the host)")
        for i in rej:
            out[i].copy_(host_window(files[i], (win_h, win_w)), non_blocking=True)
    if st is not None:
        bad = torch.nonzero(st != 0).flatten().tolist()
        bad = [i for i in bad if i not in rej]
        if bad:
            raise HcirError(f"corrupt entropy-coded data in files {bad}")
    return out
