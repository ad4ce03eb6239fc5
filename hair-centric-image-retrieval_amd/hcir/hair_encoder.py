"""hcir.hair_encoder — the retrieval front-end of src/models/hair_encoder.py on the HIP path.

  HairEncoder(ckpt_path, model_name="vit_base_patch16", device=None)   :22-41
  .extract_features(images) -> CLS of forward_features (no final norm)   :89-101,208-212
  .extract_dataset_features(data_path, batch_size, num_workers, save_dir)   :103-142
  .encode_single_image(image_path)                                       :165-178
  .retrieve_similar_images(query_embedding, all_embeddings, all_paths, top_k=5)  :180-198
  .load_embeddings / .check_embeddings_exist  (embeddings.npy + image_paths.txt)  :144-163

Image pipeline (:43-50): Resize(224, bicubic) -> CenterCrop(224) -> ToTensor -> Normalize, all on the device:
the loader workers only READ the files; PNG and baseline-JPEG files are decoded whole by hcir_png_decode_window_u8 /
hcir_jpeg_decode_window_u8 (window = image), hcir_resize_crop_bicubic_u8 applies Pillow's bicubic resize (byte-exact)
and the centre crop, hcir_knn_transform_u8 ToTensor + Normalize (bit-identical to the torchvision arithmetic).
A file outside the device decoders' subsets (bmp, webp, progressive JPEG, 16-bit PNG ...) is decoded by PIL on the
host, as the reference does, and joins the batch as RGB8 in front of the device resize.
`ImageFolder` restates torchvision.datasets.ImageFolder's directory contract (class sub-folders in sorted
order, files in sorted order, its extension list); torchvision is not a dependency.

cosine_similarity([q], G) + argsort[::-1][:top_k] becomes one hcir_sim_topk call with both
inverse norms folded in (embeddings on this path are NOT pre-normalised, :196 NOTE in
SURVEY.md §3.2).  Tie-break differs from the reference on EXACT ties only: the reference's
argsort()[::-1] yields the highest index first, this build the lowest.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

from . import models_vit, ops
from .transform import center_window_u8, knn_transform, knn_transform_u8

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


class ImageFolder:
    """torchvision.datasets.ImageFolder's contract: root/<class>/<...>/<image>; classes = sorted sub-folder
    names; samples = [(path, class_index)] walking each class folder in sorted order."""

    def __init__(self, root: str, transform=None):
        self.root = root
        self.transform = transform
        self.classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for d, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(d, f), self.class_to_idx[c]))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        from PIL import Image
        path, target = self.samples[i]
        with Image.open(path) as im:
            im = im.convert("RGB")
        return (self.transform(im) if self.transform else im), target


def resize_shorter_side(image, size: int = 224):
    """torchvision transforms.Resize(size, interpolation=3) on a PIL image: the shorter side becomes `size`,
    the other int(size * long / short), PIL bicubic (src/models/hair_encoder.py:46)."""
    from PIL import Image
    w, h = image.size
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return image
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    return image.resize((nw, nh), Image.BICUBIC)


class _FileBytes:
    """Dataset of raw file contents (uint8 tensors) in ImageFolder order: what a loader worker does is read()."""

    def __init__(self, samples):
        self.samples = samples

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        return torch.from_numpy(np.fromfile(self.samples[i][0], dtype=np.uint8))


def _identity(items):
    return items


class FeatureExtractor:
    def __init__(self, model):
        self.model = model
        self.model.eval()

    def extract_features(self, x):
        with torch.no_grad():
            return self.model.forward_features(x)[:, 0]  # CLS token


def _sniff(a: np.ndarray):
    """(codec, height, width) from the first bytes of a file, without a decoder: PNG (signature + IHDR at its fixed
    place), JPEG (marker walk to the frame header); anything else - or anything odd - goes to the host decoder, which
    also reports what is wrong with it.  Only the grouping depends on this; the stagers parse the files themselves."""
    n = a.size
    if n >= 26 and a[:8].tobytes() == b"\x89PNG\r\n\x1a\n" and a[12:16].tobytes() == b"IHDR":
        w, h = int.from_bytes(a[16:20].tobytes(), "big"), int.from_bytes(a[20:24].tobytes(), "big")
        return ("png", h, w)
    if n >= 4 and a[0] == 0xFF and a[1] == 0xD8:
        i = 2
        while i + 9 < n and a[i] == 0xFF:
            m = int(a[i + 1])
            if m == 0xFF:
                i += 1
                continue
            if 0xD0 <= m <= 0xD9 or m == 0x01:
                break
            ln = (int(a[i + 2]) << 8) | int(a[i + 3])
            if m in (0xC0, 0xC1, 0xC2):
                return ("jpeg", (int(a[i + 5]) << 8) | int(a[i + 6]), (int(a[i + 7]) << 8) | int(a[i + 8]))
            if m == 0xDA or ln < 2:
                break
            i += 2 + ln
    return ("host", 0, 0)


class HairEncoder:
    def __init__(self, ckpt_path: Optional[str], model_name: str = "vit_base_patch16", device=None):
        self.ckpt_path = ckpt_path
        self.model_name = model_name
        self.device = device if device else ("cuda" if torch.cuda.is_available() else "cpu")
        self.model = self._build_model()
        if ckpt_path is not None:
            checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            msg = self.model.load_state_dict(checkpoint["model"], strict=False)
            print("Model loading message:", msg)
        self.model.to(self.device)
        self.model.eval()
        self.feature_extractor = FeatureExtractor(self.model)
        self._stage_ring = {}   # device_windows: recycled pinned staging blobs per codec
        self._gallery_key = None
        self._gallery = None
        self._gallery_ref = None

    def _build_model(self):
        return models_vit.__dict__[self.model_name](drop_path_rate=0.1, global_pool=True, init_values=None)

    def extract_features(self, images: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self.feature_extractor.extract_features(images)

    # ---- embedding store (same on-disk format as the reference) ----
    def save_embeddings(self, all_embeddings: np.ndarray, all_paths: List[str], save_dir: str) -> None:
        os.makedirs(save_dir, exist_ok=True)
        np.save(os.path.join(save_dir, "embeddings.npy"), all_embeddings)
        with open(os.path.join(save_dir, "image_paths.txt"), "w") as f:
            for path in all_paths:
                f.write(path + "\n")

    def load_embeddings(self, save_dir):
        embeddings = np.load(os.path.join(save_dir, "embeddings.npy"))
        with open(os.path.join(save_dir, "image_paths.txt"), "r") as f:
            paths = [line.strip() for line in f.readlines()]
        return embeddings, paths

    def check_embeddings_exist(self, save_dir):
        return (os.path.exists(os.path.join(save_dir, "embeddings.npy"))
                and os.path.exists(os.path.join(save_dir, "image_paths.txt")))

    # ---- image pipeline (:43-50) ----
    def _get_transform(self):
        """Host form with the reference's signature: PIL -> fp32 [3,224,224]."""
        return lambda im: knn_transform(resize_shorter_side(im, 224), 224)

    @property
    def transform(self):
        return self._get_transform()

    @staticmethod
    def _window_u8(im) -> torch.Tensor:
        """Host form of the window (PIL resize + crop): what the device path is tested against."""
        return center_window_u8(resize_shorter_side(im, 224), 224)

    def device_windows(self, files) -> torch.Tensor:
        """Compressed files (bytes / uint8 arrays) -> the transform's RGB8 windows [B, 224, 224, 3] on the device:
        whole-image device decode per (codec, size) group, then ONE device resize + crop over the batch."""
        import io
        from PIL import Image
        from . import jpeg, png, resize
        if not str(self.device).startswith("cuda"):
            raise RuntimeError("HairEncoder.device_windows runs on a HIP device only (no CPU fallback)")
        raw = [jpeg._as_u8(f) for f in files]
        groups = {}
        for i, a in enumerate(raw):
            groups.setdefault(_sniff(a), []).append(i)
        images = [None] * len(raw)
        threads = min(16, os.cpu_count() or 8)
        for (kind, h, w), idx in groups.items():
            sub = [raw[i] for i in idx]
            whole = None
            host = list(range(len(sub)))
            if kind != "host":
                mod = png if kind == "png" else jpeg
                # the staging blobs are recycled from batch to batch, two per codec in turn (pinned: no allocation, no
                # page faults); a blob is staged into again only after the copy that read it last has finished
                ring = self._stage_ring.setdefault(kind, {"blob": [None, None], "copied": [None, None], "turn": 0})
                t = ring["turn"]
                ring["turn"] ^= 1
                if ring["copied"][t] is not None:
                    ring["copied"][t].synchronize()
                staged = mod.stage_batch(sub, threads=min(threads, len(sub)), out=ring["blob"][t])
                if ring["blob"][t] is None or staged.blob.data_ptr() != ring["blob"][t].data_ptr():
                    # first batch, or one that did not fit: a larger blob for the next turn of this slot
                    ring["blob"][t] = torch.empty(staged.blob.numel() * 5 // 4 + 65536, dtype=torch.uint8,
                                                  pin_memory=torch.cuda.is_available())
                host = staged.rejected
                if len(host) < len(sub):
                    on_dev = staged.to(self.device)
                    ring["copied"][t] = torch.cuda.Event()
                    ring["copied"][t].record(torch.cuda.current_stream(self.device))
                    whole = mod.decode_windows(on_dev, (h, w), _skip_rejected_check=True)
            for k, i in enumerate(idx):
                if k in host:  # the reference's own decoder for this file
                    with Image.open(io.BytesIO(sub[k])) as im:
                        images[i] = torch.from_numpy(np.asarray(im.convert("RGB")).copy()).to(self.device)
                else:
                    images[i] = whole[k]
        return resize.resize_center_crop(images, 224)

    def extract_dataset_features(self, data_path, batch_size=64, num_workers=8, save_dir="embeddings"):
        """Embed every image of an ImageFolder tree, save embeddings.npy + image_paths.txt (:103-142).
        DataLoader workers read the files; decode, Resize, CenterCrop, ToTensor, Normalize and the ViT run on the
        device; embeddings stay in HBM until the single copy back at the end."""
        from torch.utils.data import DataLoader
        print(f"Loading dataset from: {data_path}")
        dataset = ImageFolder(data_path)
        loader = DataLoader(_FileBytes(dataset.samples), batch_size=batch_size, shuffle=False, num_workers=num_workers,
                            collate_fn=_identity)
        feats, all_paths = [], []
        with torch.no_grad():
            for files in loader:
                x = knn_transform_u8(self.device_windows(files))
                feats.append(self.extract_features(x).float().clone())   # the engine re-uses its buffers
                start_idx = len(all_paths)
                all_paths.extend([dataset.samples[i][0] for i in range(start_idx, start_idx + len(files))])
        all_embeddings = torch.cat(feats, 0).cpu().numpy()
        self.save_embeddings(all_embeddings, all_paths, save_dir)
        print(f"Saved {all_embeddings.shape[0]} embeddings and paths to {save_dir}")
        return all_embeddings, all_paths

    def encode_single_image(self, image_path):
        """One image -> embedding (np.ndarray [D]) (:165-178)."""
        x = knn_transform_u8(self.device_windows([np.fromfile(image_path, dtype=np.uint8)]))
        return self.extract_features(x).float().cpu().numpy()[0]

    # ---- retrieval ----
    def set_gallery(self, all_embeddings) -> None:
        """Upload a gallery (and its inverse norms) into HBM.  retrieve_similar_images does this itself;
        call it again after editing the array IN PLACE (the cache cannot see that)."""
        g = torch.as_tensor(np.ascontiguousarray(all_embeddings, dtype=np.float32)).to(self.device)
        self._gallery = (g, ops.row_invnorm(g, 1e-12))
        self._gallery_ref = all_embeddings            # strong reference: the id cannot be recycled
        self._gallery_key = self._fingerprint(all_embeddings)

    @staticmethod
    def _fingerprint(a):
        """Cheap content check of the cached array: buffer address, shape and a checksum of a few rows."""
        a = np.asarray(a)
        n = a.shape[0]
        rows = sorted({0, n // 3, (2 * n) // 3, n - 1}) if n else []
        ptr = a.__array_interface__["data"][0]
        return (ptr, a.shape, str(a.dtype), tuple(float(np.asarray(a[r], dtype=np.float64).sum()) for r in rows))

    def invalidate_gallery(self) -> None:
        self._gallery = self._gallery_ref = self._gallery_key = None

    def _resident(self, all_embeddings):
        """The gallery resident in HBM; re-uploaded when another array (or changed content) is passed.
        The reference recomputes against the passed array on every call (:193)."""
        if getattr(self, "_gallery_ref", None) is not all_embeddings \
                or self._gallery_key != self._fingerprint(all_embeddings):
            self.set_gallery(all_embeddings)
        return self._gallery

    def retrieve_similar_images(self, query_embedding, all_embeddings, all_paths, top_k=5):
        g, gn = self._resident(all_embeddings)
        q = torch.as_tensor(np.ascontiguousarray(query_embedding, dtype=np.float32)).reshape(1, -1).to(self.device)
        top_k = min(top_k, g.shape[0])  # numpy slicing [:top_k] never over-runs
        val, idx = ops.sim_topk(q, g, top_k, q_inv_norm=ops.row_invnorm(q, 1e-12), g_inv_norm=gn)
        val, idx = val[0].cpu().numpy(), idx[0].cpu().numpy()
        return [{"path": all_paths[i], "similarity": v} for v, i in zip(val, idx)]
