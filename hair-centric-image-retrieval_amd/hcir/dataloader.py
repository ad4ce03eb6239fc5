"""hcir.dataloader — the reference's annotation-CSV dataset contract, host side only.

Contract kept (HP/utils/dataloader.py:13-41): `CustomDataset(annotations_file, img_dir,
transform=None, our_method=False)`; the CSV's first column is a file name relative to
`img_dir`, the second an integer class; an item is `(transform(image), label)`, or the dict
`{"anchor", "pos1"}` of a two-view transform when `our_method` is set.  Decoding uses PIL
(torchvision.io is not a dependency).  Not on the device hot path: it exists so that the
kNN CLI runs end to end on real folders.
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
