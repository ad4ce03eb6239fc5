"""hcir — MI355X-native retrieval hot path (Python host side).

Mirrors the reference's interfaces for the hot path only:
  hcir.main_backbone  SHAM2, ViTWrapper          (HP/src/main_backbone.py:528-637)
  hcir.models_vit     VisionTransformer & ctors  (HP/src/models_vit.py)
  hcir.backbone       SimCLR, MAE signatures     (HP/src/backbone.py:462-525,648-681)
  hcir.classification_engine  Classifier.knn_eval (HP/src/classification_engine.py:39-98)
  hcir.neg_sampling   NegSamplerStatic           (HP/src/neg_sampling.py:26-53)
  hcir.hair_encoder   retrieve_similar_images    (src/models/hair_encoder.py:180-198)
  hcir.losses         NTXentLoss                 (lightly; HP/src/pretrain_engine.py:93,725)
Compute goes through hcir.ops -> libhcir.so (HIP, gfx950).  No CPU fallback.
"""
from ._lib import HcirError, build, lib  # noqa: F401
