"""hcir.dataloader — CustomDataset with the reference's CSV contract (HP/utils/dataloader.py:13-41):
annotations `id,class`; item -> (transform(image), label).  Host-side I/O (PIL), not on the
device hot path; present so the CLI runs end to end."""
from __future__ import annotations

import os

import pandas as pd
from PIL import Image
from torch.utils.data import Dataset


class CustomDataset(Dataset):
    def __init__(self, annotations_file, img_dir, transform=None, our_method=False):
        self.img_labels = pd.read_csv(annotations_file)
        self.img_dir = img_dir
        self.transform = transform
        self.our_method = our_method

    def __len__(self):
        return len(self.img_labels)

    def __getitem__(self, idx):
        img_name = self.img_labels.iloc[idx, 0]
        label = self.img_labels.iloc[idx, 1]
        image = Image.open(os.path.join(self.img_dir, img_name)).convert("RGB")
        if self.our_method:
            anchor, pos1 = self.transform(image)
            return {"anchor": anchor, "pos1": pos1}
        return self.transform(image), label
