"""hcir.transform — knn_transform (HP/utils/transform.py:10-14) without torchvision:
CenterCrop(224) -> ToTensor -> Normalize(ImageNet mean/std).  No resize (the resize variant is
commented out in the reference, :15-19).  Host-side input contract of the embed path."""
from __future__ import annotations

import numpy as np
import torch

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def knn_transform(image, size: int = 224) -> torch.Tensor:
    """PIL image -> float32 tensor [3, size, size]."""
    arr = np.asarray(image.convert("RGB"))
    h, w = arr.shape[:2]
    if h < size or w < size:  # torchvision CenterCrop pads smaller images with zeros
        ph, pw = max(size - h, 0), max(size - w, 0)
        arr = np.pad(arr, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        h, w = arr.shape[:2]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    win = arr[top:top + size, left:left + size].astype(np.float32) / np.float32(255.0)
    win = (win - _MEAN) / _STD
    return torch.from_numpy(np.ascontiguousarray(win.transpose(2, 0, 1)))
