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


def positive_masking(images: np.ndarray, u: np.ndarray, keys: np.ndarray, patch_size: int = 32,
                     threshold: float = 0.01):
    """PositiveMaskingTransform.__call__ (HP/utils/transform.py:101-150) restated as the same per-image loop,
    with its randomness as data: u[b] is the mask ratio the reference draws with uniform_ (:133), and the random
    subset it takes with randperm (:139) is the `num_mask` hair patches with the smallest keys[b] (ties: smaller
    patch index).  images float32 [B, C, H, W].  Returns (masked images, number of zeroed patches per image)."""
    B, C, H, W = images.shape
    nh, nw = H // patch_size, W // patch_size
    out = images.copy()
    counts = np.zeros(B, dtype=np.int32)
    for b in range(B):
        means = np.zeros(nh * nw, dtype=np.float64)
        for ph in range(nh):
            for pw in range(nw):
                blk = images[b, :, ph * patch_size:(ph + 1) * patch_size, pw * patch_size:(pw + 1) * patch_size]
                means[ph * nw + pw] = blk.astype(np.float64).mean()            # :124 mean over (C, ph, pw)
        hair = np.nonzero(means > threshold)[0]                               # :125,130
        if len(hair) == 0:
            continue
        num_mask = int(len(hair) * float(u[b]))                               # :134-135
        if num_mask == 0:
            continue
        order = sorted(hair, key=lambda i: (float(keys[b, i]), i))
        for idx in order[:num_mask]:                                          # :142-146
            ph, pw = idx // nw, idx % nw
            out[b, :, ph * patch_size:(ph + 1) * patch_size, pw * patch_size:(pw + 1) * patch_size] = 0.0
        counts[b] = num_mask
    return out, counts


def positive_transform(images, angle: float, sigma: float):
    """`positive_transform` (HP/utils/transform.py:21-24) on a batch tensor, restated with torch CPU ops from
    torchvision's PUBLIC source (torchvision is not installed here: PARITY UNPINNED for this function):
      RandomRotation -> F.rotate(img, angle, NEAREST, expand=False, center=None, fill=0):
        theta = _get_inverse_affine_matrix([0, 0], -angle, [0, 0], 1, [0, 0]) = [[cos r, sin r, 0], [-sin r, cos r, 0]],
        r = radians(-angle); grid = _gen_affine_grid (pixel centres, theta^T / (w/2, h/2));
        grid_sample(nearest, zeros, align_corners=False)
      GaussianBlur(3, sigma) -> reflect-pad by 1, depthwise conv2d with outer(k1d, k1d), k1d = normalised
        exp(-0.5 (x / sigma)^2) on x = linspace(-1, 1, 3)
    images: float32 torch tensor [B, C, H, W] (CPU)."""
    import math
    import torch
    import torch.nn.functional as F
    b, c, h, w = images.shape
    rot = math.radians(-angle)
    theta = torch.tensor([[math.cos(rot), math.sin(rot), 0.0], [-math.sin(rot), math.cos(rot), 0.0]],
                         dtype=torch.float32).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).unsqueeze(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2).expand(b, h, w, 2)
    rotated = F.grid_sample(images, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    x = torch.linspace(-1.0, 1.0, steps=3, dtype=torch.float32)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :]).expand(c, 1, 3, 3)
    padded = F.pad(rotated, [1, 1, 1, 1], mode="reflect")
    return F.conv2d(padded, k2, groups=c)
