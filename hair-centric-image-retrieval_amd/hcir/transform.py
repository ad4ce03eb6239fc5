"""hcir.transform — knn_transform (HP/utils/transform.py:10-14) without torchvision:
CenterCrop(224) -> ToTensor -> Normalize(ImageNet mean/std).  No resize (the resize variant is
commented out in the reference, :15-19).

`knn_transform(pil)` is the host form (PIL + numpy) with the reference's signature;
`knn_transform_u8(batch)` runs the same arithmetic on the HIP device over decoded RGB8 images
(hcir_knn_transform_u8): the loader ships bytes (4x less H2D traffic than fp32) and the float work
leaves the DataLoader workers."""
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


def knn_transform_u8(images: torch.Tensor, size: int = 224) -> torch.Tensor:
    """uint8 [B, H, W, 3] (or [H, W, 3]) on the HIP device -> float32 [B, 3, size, size]; bit-identical to
    knn_transform applied image by image.  No CPU path."""
    from . import _lib
    from ._lib import HcirError, check
    import ctypes
    if images.dtype != torch.uint8:
        raise HcirError(f"knn_transform_u8 takes uint8 RGB images, got {images.dtype}")
    if not images.is_cuda:
        raise HcirError(f"images are on {images.device}; knn_transform_u8 runs on a HIP device only")
    if images.dim() == 3:
        images = images[None]
    if images.dim() != 4 or images.shape[-1] != 3:
        raise HcirError(f"expected [B, H, W, 3], got {tuple(images.shape)}")
    images = images.contiguous()
    b, h, w, _ = images.shape
    out = torch.empty((b, 3, size, size), dtype=torch.float32, device=images.device)
    mean = (ctypes.c_float * 3)(*_MEAN.tolist())
    std = (ctypes.c_float * 3)(*_STD.tolist())
    check(_lib.lib().hcir_knn_transform_u8(images.data_ptr(), b, h, w, size, mean, std, out.data_ptr(),
                                           torch.cuda.current_stream(images.device).cuda_stream),
          "hcir_knn_transform_u8")
    return out


def center_window_u8(image, size: int = 224) -> torch.Tensor:
    """PIL image -> uint8 [size, size, 3]: the CenterCrop window only (zero-padded when the image is
    smaller), for loaders that leave ToTensor + Normalize to knn_transform_u8 on the device."""
    arr = np.asarray(image.convert("RGB"))
    h, w = arr.shape[:2]
    if h < size or w < size:
        ph, pw = max(size - h, 0), max(size - w, 0)
        arr = np.pad(arr, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        h, w = arr.shape[:2]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return torch.from_numpy(np.ascontiguousarray(arr[top:top + size, left:left + size]))


class PositiveMaskingTransform:
    """HP/utils/transform.py:84-150 on the HIP device (hcir_positive_masking): same constructor and call,
    no per-image host round trip.  `__call__(images, generator=None)` draws the per-image mask ratio and one
    key per patch with torch's generator on the device (the reference's uniform_ + randperm), or takes them
    explicitly through `apply(images, u, keys)` — which is what the parity tests drive."""

    def __init__(self, patch_size=32, mask_ratio_range=(0.1, 0.2), threshold=0.01):
        self.patch_size = patch_size
        self.mask_ratio_range = mask_ratio_range
        self.threshold = threshold

    def apply(self, images: torch.Tensor, u: torch.Tensor, keys: torch.Tensor, return_counts: bool = False):
        from . import _lib
        from ._lib import HcirError, check
        if not isinstance(images, torch.Tensor):
            raise ValueError("Input must be a torch.Tensor")
        if not images.is_cuda:
            raise HcirError(f"images are on {images.device}; PositiveMaskingTransform runs on a HIP device only")
        if images.dim() != 4:
            raise HcirError(f"expected [B, C, H, W], got {tuple(images.shape)}")
        x = images.float().contiguous()
        b, c, h, w = x.shape
        p = int(self.patch_size)
        npatch = (h // p) * (w // p)
        u = u.to(device=x.device, dtype=torch.float32).contiguous()
        keys = keys.to(device=x.device, dtype=torch.float32).contiguous()
        if u.numel() != b or keys.numel() != b * npatch:
            raise HcirError("u must have B entries and keys B * (H/patch) * (W/patch)")
        out = torch.empty_like(x)
        cnt = torch.empty(b, dtype=torch.int32, device=x.device)
        check(_lib.lib().hcir_positive_masking(x.data_ptr(), b, c, h, w, p, float(self.threshold), u.data_ptr(),
                                               keys.data_ptr(), out.data_ptr(), cnt.data_ptr(),
                                               torch.cuda.current_stream(x.device).cuda_stream),
              "hcir_positive_masking")
        return (out, cnt) if return_counts else out

    def __call__(self, images: torch.Tensor, generator=None) -> torch.Tensor:
        if not isinstance(images, torch.Tensor):
            raise ValueError("Input must be a torch.Tensor")
        b, _, h, w = images.shape
        npatch = (h // self.patch_size) * (w // self.patch_size)
        lo, hi = self.mask_ratio_range
        u = torch.empty(b, device=images.device).uniform_(lo, hi, generator=generator)
        keys = torch.rand((b, npatch), device=images.device, generator=generator)
        return self.apply(images, u, keys)


class PositiveTransform:
    """`positive_transform` of HP/utils/transform.py:21-24 — T.Compose([T.RandomRotation((-15, 15)),
    T.GaussianBlur(kernel_size=3, sigma=(0.1, 0.5))]) — as the step applies it to the DEVICE batch
    (HP/src/pretrain_engine.py:686), on the HIP path (hcir_positive_transform: one kernel for both).

    torchvision draws ONE angle and ONE sigma per call for a batch tensor, from torch's CPU generator
    (`torch.empty(1).uniform_(lo, hi)`, rotation first, then blur): `__call__` draws them the same way, so a seeded
    run takes the same augmentation parameters as the reference; `apply(images, angle, sigma)` takes them explicitly
    (what the parity test drives)."""

    def __init__(self, degrees=(-15.0, 15.0), kernel_size: int = 3, sigma=(0.1, 0.5)):
        if kernel_size != 3:
            raise ValueError("hcir_positive_transform implements the reference's kernel_size=3")
        self.degrees = (float(degrees[0]), float(degrees[1]))
        self.sigma = (float(sigma[0]), float(sigma[1]))

    @staticmethod
    def params_to_host_arrays(angle: float, sigma: float):
        """(theta4, taps2) as torchvision derives them: F.rotate -> _get_inverse_affine_matrix([0, 0], -angle, ...),
        _get_gaussian_kernel1d(3, sigma) in float32."""
        import ctypes
        import math
        rot = math.radians(-angle)
        theta = torch.tensor([math.cos(rot), math.sin(rot), -math.sin(rot), math.cos(rot)], dtype=torch.float32)
        x = torch.linspace(-1.0, 1.0, steps=3, dtype=torch.float32)
        pdf = torch.exp(-0.5 * (x / sigma).pow(2))
        k = pdf / pdf.sum()
        return (ctypes.c_float * 4)(*theta.tolist()), (ctypes.c_float * 2)(float(k[0]), float(k[1]))

    def apply(self, images: torch.Tensor, angle: float, sigma: float) -> torch.Tensor:
        from . import _lib
        from ._lib import HcirError, check
        if not isinstance(images, torch.Tensor) or images.dim() != 4:
            raise HcirError("positive_transform takes a [B, C, H, W] tensor")
        if not images.is_cuda:
            raise HcirError(f"images are on {images.device}; PositiveTransform runs on a HIP device only")
        x = images.float().contiguous()
        b, c, h, w = x.shape
        out = torch.empty_like(x)
        theta4, taps2 = self.params_to_host_arrays(float(angle), float(sigma))
        check(_lib.lib().hcir_positive_transform(x.data_ptr(), b, c, h, w, theta4, taps2, out.data_ptr(),
                                                 torch.cuda.current_stream(x.device).cuda_stream),
              "hcir_positive_transform")
        return out

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        angle = float(torch.empty(1).uniform_(self.degrees[0], self.degrees[1]).item())   # RandomRotation.get_params
        sigma = float(torch.empty(1).uniform_(self.sigma[0], self.sigma[1]).item())        # GaussianBlur.get_params
        return self.apply(images, angle, sigma)


positive_transform = PositiveTransform()
