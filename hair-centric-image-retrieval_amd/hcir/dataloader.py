"""hcir.dataloader — the reference's annotation-CSV dataset contract, host side only.

Contract kept (HP/utils/dataloader.py:13-41): `CustomDataset(annotations_file, img_dir,
transform=None, our_method=False)`; the CSV's first column is a file name relative to
`img_dir`, the second an integer class; an item is `(transform(image), label)`, or the dict
`{"anchor", "pos1"}` of a two-view transform when `our_method` is set.  Decoding uses PIL
(torchvision.io is not a dependency).

`EncodedDataset` + `collate_encoded` are the device-decode form of the same contract: the
workers only READ the files and stage them (`hcir.jpeg.stage_batch`: marker walk + unstuffing copy into
one blob); the batch that reaches the engine is a `StagedBatch`, which `Classifier._embed` uploads and decodes
with `hcir_png_decode_window_u8` (the reference's `*_hair.png` crops, `HP/data/data_train.csv`) or
`hcir_jpeg_decode_window_u8` (its `*_full_face` JPEG lists) -> `hcir_knn_transform_u8`.  Files outside both device
subsets (progressive JPEG, 16-bit or interlaced PNG ...) are decoded by the worker with PIL, as the reference does,
and ride along as RGB8 windows.
"""
from __future__ import annotations

import csv
from pathlib import Path
from typing import Callable, List, Optional, Tuple

from PIL import Image
from torch.utils.data import Dataset


def _read_annotations(path: str) -> List[Tuple[str, int]]:
    """[(file name, class id)] from a header-ed two-column CSV (`id,class`)."""
    records: List[Tuple[str, int]] = []
    with open(path, newline="") as handle:
        reader = csv.reader(handle)
        next(reader, None)  # header row
        for row in reader:
            if len(row) >= 2 and row[0]:
                records.append((row[0], int(float(row[1]))))
    return records


class CustomDataset(Dataset):
    def __init__(self, annotations_file, img_dir, transform: Optional[Callable] = None, our_method=False):
        self.records = _read_annotations(annotations_file)
        self.root = Path(img_dir)
        self.transform = transform
        self.our_method = bool(our_method)

    def __len__(self) -> int:
        return len(self.records)

    def _decode(self, name: str) -> Image.Image:
        with Image.open(self.root / name) as im:
            return im.convert("RGB")

    def __getitem__(self, index: int):
        name, label = self.records[index]
        views = self.transform(self._decode(name)) if self.transform else self._decode(name)
        if self.our_method:
            first, second = views
            return {"anchor": first, "pos1": second}
        return views, label


class EncodedDataset(Dataset):
    """Same CSV contract as CustomDataset; an item is (file bytes as a uint8 tensor, label)."""

    def __init__(self, annotations_file, img_dir):
        self.records = _read_annotations(annotations_file)
        self.root = Path(img_dir)

    def __len__(self) -> int:
        return len(self.records)

    def __getitem__(self, index: int):
        import numpy as np
        import torch
        name, label = self.records[index]
        return torch.from_numpy(np.fromfile(self.root / name, dtype=np.uint8)), label


class EncodedBatch:
    """What collate_encoded hands to the engine: one staging blob per codec (JPEG files, PNG files) with the batch
    positions they fill, plus host-decoded windows of the files neither device decoder takes."""

    def __init__(self, parts, host_windows, size, n):
        self.parts, self.host_windows, self.size, self.n = parts, host_windows, size, n

    @property
    def staged(self):  # the single-codec batch's blob (kept for callers that look at sizes / bytes)
        return self.parts[0][1] if len(self.parts) == 1 else None

    def decode(self, device, check_status: bool = True):
        """-> uint8 [B, size, size, 3] on `device` (device decode; host-decoded files' windows copied in).
        check_status: read the decoders' per-image status back (one 4 B/image D2H per batch) and raise on a corrupt
        stream, as the reference's decode_image / PIL raise for the same file."""
        import torch
        from . import jpeg, png
        from ._lib import HcirError
        out = None
        for kind, staged, idx in self.parts:
            mod = jpeg if kind == "jpeg" else png
            try:
                win = mod.decode_windows(staged.to(device), self.size, check_status=check_status,
                                         _skip_rejected_check=True)
            except HcirError as e:
                raise HcirError(f"{e} (batch positions of this codec's files: {idx})") from None
            if len(self.parts) == 1 and len(idx) == self.n:
                out = win
            else:
                if out is None:
                    out = torch.zeros((self.n, self.size, self.size, 3), dtype=torch.uint8, device=device)
                out[torch.as_tensor(idx, device=device)] = win
        if out is None:
            out = torch.zeros((self.n, self.size, self.size, 3), dtype=torch.uint8, device=device)
        for i, win in self.host_windows.items():
            out[i].copy_(win, non_blocking=True)
        return out


def collate_encoded(items, size: int = 224, verify_crc: bool = True):
    """DataLoader collate_fn for EncodedDataset: (EncodedBatch, labels).  PNG files (the reference's *_hair.png
    crops) and baseline JPEGs are staged for their device decoders; anything else is decoded here with PIL."""
    import torch
    from . import jpeg, png
    files = [it[0] for it in items]
    is_png = [png.is_png(f) for f in files]
    parts, host = [], {}
    for kind, mod, sel in (("png", png, [i for i, p in enumerate(is_png) if p]),
                           ("jpeg", jpeg, [i for i, p in enumerate(is_png) if not p])):
        if not sel:
            continue
        sub = [files[i] for i in sel]
        # a worker is one process: its own core
        staged = mod.stage_batch(sub, pin=False, threads=1, verify_crc=verify_crc) if kind == "png" else \
            mod.stage_batch(sub, pin=False, threads=1)
        for j in staged.rejected:
            host[sel[j]] = jpeg.host_window(sub[j], (size, size))
        if len(staged.rejected) < len(sub):
            parts.append((kind, staged, sel))
    return EncodedBatch(parts, host, size, len(files)), torch.as_tensor([int(it[1]) for it in items])
