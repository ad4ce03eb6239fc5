"""hcir.png — PNG decode of CenterCrop windows on the HIP device (include/hcir.h, csrc/png.hip).

The format of every hair-region crop the reference lists (`HairPretraining/data/data_train.csv`: `*_hair.png`;
`assets/hair_region_only/*.png`).  Stands where the reference decodes on the host in front of `knn_transform`:
`read_file` + `torchvision.io.decode_image(img_bytes, mode=RGB)` (HP/utils/dataloader.py:28-31) and
`Image.open(path).convert('RGB')` (src/models/hair_encoder.py:108,169).  inflate and the scanline filters are exact
integer algorithms, so the output bytes equal libpng's / Pillow's.

    staged = stage_batch([bytes, ...])            # host: chunk walk (+ CRC-32) + IDAT concatenation into ONE blob
    win    = decode_windows(staged.to(device))    # device: [B, 224, 224, 3] uint8, the CenterCrop(224) windows

Files outside the 8-bit non-interlaced subset (16-bit, 1/2/4-bit, Adam7) are reported in `staged.rejected`;
`decode_windows` fills their windows from `hcir.jpeg.host_window(...)` (PIL, as the reference does) when asked to,
else raises.  There is no CPU path for the supported files: `decode_windows` refuses a non-HIP blob.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import HcirError, check
from .jpeg import Bytes, _as_u8, host_window

VERIFY_CRC = 1
PNG_MAGIC = b"\x89PNG\r\n\x1a\n"


class PngHeader(ctypes.Structure):
    """Mirror of hcir_png_header (include/hcir.h)."""
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("color_type", ctypes.c_int32),
                ("bpp", ctypes.c_int32), ("stream_bytes", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("stage_offset", ctypes.c_uint64), ("palette", ctypes.c_uint8 * 768)]


HEADER_BYTES = ctypes.sizeof(PngHeader)


def is_png(buf: Bytes) -> bool:
    a = _as_u8(buf)
    return a.size >= 8 and a[:8].tobytes() == PNG_MAGIC


class StagedBatch:
    """One staging blob: `b` headers, then every image's zlib stream (the IDAT payloads concatenated)."""

    def __init__(self, blob: torch.Tensor, b: int, status: np.ndarray, host_headers: Optional[torch.Tensor] = None):
        self.blob, self.b, self.status = blob, b, status
        self._host_headers = host_headers if host_headers is not None else blob[:b * HEADER_BYTES]

    @property
    def rejected(self) -> List[int]:
        return [int(i) for i in np.nonzero(self.status != 0)[0]]

    def headers(self):
        return (PngHeader * self.b).from_address(self._host_headers.data_ptr())

    def sizes(self) -> List[Tuple[int, int]]:
        return [(h.height, h.width) for h in self.headers()]

    def stream_bytes(self) -> int:
        return int(sum(h.stream_bytes for h in self.headers()))

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
                verify_crc: bool = True, out: Optional[torch.Tensor] = None) -> StagedBatch:
    """Parse + stage every file into one blob (hcir_png_stage_batch).  Host only; needs no GPU.
    `out`: a host uint8 blob to stage into (a loader recycles its pinned blobs); a blob that is too small is replaced
    by a fresh one."""
    L = _lib.lib()
    arrs = [_as_u8(f) for f in files]
    b = len(arrs)
    if b == 0:
        raise HcirError("stage_batch needs at least one file")
    ptrs = (ctypes.c_void_p * b)(*[a.ctypes.data for a in arrs])
    lens = (ctypes.c_size_t * b)(*[a.size for a in arrs])
    status = np.zeros(b, dtype=np.int32)
    used = ctypes.c_size_t(0)
    flags = VERIFY_CRC if verify_crc else 0
    if out is not None:
        if out.is_cuda or out.dtype != torch.uint8 or not out.is_contiguous():
            raise HcirError("stage_batch(out=...) needs a contiguous host uint8 tensor")
        rc = L.hcir_png_stage_batch(ptrs, lens, b, flags, out.data_ptr(), out.numel(), ctypes.byref(used),
                                    status.ctypes.data, threads)
        if rc == 0:
            return StagedBatch(out[:used.value], b, status)
        if rc != -4:                      # anything but HCIR_ERR_WORKSPACE
            check(rc, "hcir_png_stage_batch")
    else:
        check(L.hcir_png_stage_batch(ptrs, lens, b, flags, None, 0, ctypes.byref(used), status.ctypes.data, threads),
              "hcir_png_stage_batch(size)")
    if pin is None:
        pin = torch.cuda.is_available()
    blob = torch.empty(used.value, dtype=torch.uint8, pin_memory=bool(pin))
    check(L.hcir_png_stage_batch(ptrs, lens, b, flags, blob.data_ptr(), blob.numel(), ctypes.byref(used),
                                 status.ctypes.data, threads), "hcir_png_stage_batch")
    return StagedBatch(blob, b, status)


_ws = {}


def decode_windows(staged: StagedBatch, size: Union[int, Tuple[int, int]] = 224, files: Optional[Sequence[Bytes]] = None,
                   host_fallback_for_rejected: bool = False, check_status: bool = False,
                   _skip_rejected_check: bool = False) -> torch.Tensor:
    """[B, win_h, win_w, 3] uint8 on the blob's device: CenterCrop(size) of every decoded image (zero where the
    window leaves the image, as torchvision pads).  Asynchronous on the current stream unless `check_status`."""
    if not staged.blob.is_cuda:
        raise HcirError(f"the staging blob is on {staged.blob.device}; hcir_png_decode_window_u8 runs on a HIP "
                        "device only (no CPU fallback) — call staged.to(device) first")
    win_h, win_w = (size, size) if isinstance(size, int) else size
    L = _lib.lib()
    dev = staged.blob.device
    hdrs = staged._host_headers.data_ptr()
    rej = [] if _skip_rejected_check else staged.rejected
    # a rejected file's header is zeroed: the device leaves its window alone, so start those from zeros
    out = (torch.zeros if staged.rejected else torch.empty)((staged.b, win_h, win_w, 3), dtype=torch.uint8, device=dev)
    wsb = L.hcir_png_workspace_bytes(hdrs, staged.b, win_h, win_w)
    if wsb == 0:
        raise HcirError("hcir_png_workspace_bytes: invalid headers")
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < wsb:
        ws = _ws[key] = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.empty(staged.b, dtype=torch.int32, device=dev) if check_status else None
    check(L.hcir_png_decode_window_u8(staged.blob.data_ptr(), hdrs, staged.b, win_h, win_w, out.data_ptr(),
                                      None if st is None else st.data_ptr(), ws.data_ptr(), ws.numel(), stream),
          "hcir_png_decode_window_u8")
    if rej:
        if not host_fallback_for_rejected or files is None:
            raise HcirError(f"files {rej} are outside the device decoder's PNG subset (8-bit, non-interlaced) "
                            "(pass files= and host_fallback_for_rejected=True to decode those on the host)")
        for i in rej:
            out[i].copy_(host_window(files[i], (win_h, win_w)), non_blocking=True)
    if st is not None:
        bad = [i for i in torch.nonzero(st != 0).flatten().tolist() if i not in staged.rejected]
        if bad:
            raise HcirError(f"corrupt PNG data in files {bad}")
    return out
