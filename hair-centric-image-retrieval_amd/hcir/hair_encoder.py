"""hcir.hair_encoder — the retrieval front-end of src/models/hair_encoder.py on the HIP path.

  HairEncoder(ckpt_path, model_name="vit_base_patch16", device=None)   :22-41
  .extract_features(images) -> CLS of forward_features (no final norm)   :89-101,208-212
  .retrieve_similar_images(query_embedding, all_embeddings, all_paths, top_k=5)  :180-198
  .load_embeddings / .check_embeddings_exist  (embeddings.npy + image_paths.txt)  :144-163

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


class FeatureExtractor:
    def __init__(self, model):
        self.model = model
        self.model.eval()

    def extract_features(self, x):
        with torch.no_grad():
            return self.model.forward_features(x)[:, 0]  # CLS token


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
        self._gallery_key = None
        self._gallery = None

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

    # ---- retrieval ----
    def _resident(self, all_embeddings):
        """Upload the gallery once and keep it (and its inverse norms) resident in HBM."""
        key = (id(all_embeddings), getattr(all_embeddings, "shape", None))
        if key != self._gallery_key:
            g = torch.as_tensor(np.ascontiguousarray(all_embeddings, dtype=np.float32)).to(self.device)
            self._gallery = (g, ops.row_invnorm(g, 1e-12))
            self._gallery_key = key
        return self._gallery

    def retrieve_similar_images(self, query_embedding, all_embeddings, all_paths, top_k=5):
        g, gn = self._resident(all_embeddings)
        q = torch.as_tensor(np.ascontiguousarray(query_embedding, dtype=np.float32)).reshape(1, -1).to(self.device)
        top_k = min(top_k, g.shape[0])  # numpy slicing [:top_k] never over-runs
        val, idx = ops.sim_topk(q, g, top_k, q_inv_norm=ops.row_invnorm(q, 1e-12), g_inv_norm=gn)
        val, idx = val[0].cpu().numpy(), idx[0].cpu().numpy()
        return [{"path": all_paths[i], "similarity": v} for v, i in zip(val, idx)]
