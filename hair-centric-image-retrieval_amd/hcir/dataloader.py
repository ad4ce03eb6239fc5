"""hcir.dataloader — the reference's annotation-CSV dataset contract, host side only.

Contract kept (HP/utils/dataloader.py:13-41): `CustomDataset(annotations_file, img_dir,
transform=None, our_method=False)`; the CSV's first column is a file name relative to
`img_dir`, the second an integer class; an item is `(transform(image), label)`, or the dict
`{"anchor", "pos1"}` of a two-view transform when `our_method` is set.  Decoding uses PIL
(torchvision.io is not a dependency).

`EncodedDataset` + `collate_encoded` are the device-decode form of the same contract: the
workers only READ the files and stage them (`hcir.jpeg.stage_batch`: marker walk + unstuffing copy into
one blob); the batch that reaches the engine is a `StagedBatch`, which `Classifier._embed` uploads and decodes
with `hcir_jpeg_decode_window_u8` -> `hcir_knn_transform_u8`.  Files outside the baseline-JPEG subset (PNG,
progressive ...) are decoded by the worker with PIL, as the reference does, and ride along as RGB8 windows.
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
    """What collate_encoded hands to the engine: the staging blob, plus host-decoded windows of rejected files."""

    def __init__(self, staged, host_windows, size):
        self.staged, self.host_windows, self.size = staged, host_windows, size

    def decode(self, device):
        """-> uint8 [B, size, size, 3] on `device` (device decode; rejected files' windows copied in)."""
        from . import jpeg
        staged = self.staged.to(device)
        out = jpeg.decode_windows(staged, self.size) if not staged.rejected else \
            jpeg.decode_windows(staged, self.size, files=[None] * staged.b, host_fallback_for_rejected=False,
                                _skip_rejected_check=True)
        for i, win in self.host_windows.items():
            out[i].copy_(win, non_blocking=True)
        return out


def collate_encoded(items, size: int = 224):
    """DataLoader collate_fn for EncodedDataset: (EncodedBatch, labels)."""
    import torch
    from . import jpeg
    files = [it[0] for it in items]
    staged = jpeg.stage_batch(files, pin=False, threads=1)  # a worker is one process: its own core
    host = {i: jpeg.host_window(files[i], (size, size)) for i in staged.rejected}
    return EncodedBatch(staged, host, size), torch.as_tensor([int(it[1]) for it in items])
