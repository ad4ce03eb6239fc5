"""oracle.transform — knn_transform restated with PIL + numpy (TEST INFRASTRUCTURE ONLY).

HP/utils/transform.py:10-14: CenterCrop(224) -> ToTensor (/255, HWC->CHW) ->
Normalize(mean [.485,.456,.406], std [.229,.224,.225]).  NO resize (the resize variant
is commented out, :15-19): on the 1024^2 assets this is the centre 224^2 window.
torchvision CenterCrop: top = round((H - 224) / 2), left = round((W - 224) / 2).
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def window_to_tensor(rgb_u8: np.ndarray) -> np.ndarray:
    """uint8 HWC window -> normalised float32 CHW (ToTensor + Normalize)."""
    x = rgb_u8.astype(np.float32) / np.float32(255.0)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def knn_transform(pil_image, size: int = 224) -> np.ndarray:
    img = np.asarray(pil_image.convert("RGB"))
    h, w = img.shape[:2]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return window_to_tensor(img[top:top + size, left:left + size])


def grid_windows(pil_image, n: int, size: int = 224):
    """n deterministic 224^2 windows on a regular grid (config C1's extra crops)."""
    img = np.asarray(pil_image.convert("RGB"))
    h, w = img.shape[:2]
    per = int(np.ceil(np.sqrt(n)))
    ys = np.linspace(0, h - size, per).astype(int)
    xs = np.linspace(0, w - size, per).astype(int)
    out = []
    for y in ys:
        for x in xs:
            if len(out) < n:
                out.append(window_to_tensor(img[y:y + size, x:x + size]))
    return np.stack(out)
