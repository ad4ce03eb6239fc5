"""hcir.pretrain_engine — the HSimCLR training step of Trainer.train_one_epoch_SHAM on the MI355X hot path
(HP/src/pretrain_engine.py:602-757).

`SHAMTrainStep` is the body of the reference's batch loop (:618-751) with every hot operation on the HIP path:

    update_momentum(backbone / head -> momentum twins)        hcir.momentum         one launch (hcir_ema_update)
    NegSamplerRandomly / NegSamplerStatic                      hcir.neg_sampling     hcir_sim_topk
    model(neg), model(pos), model(anchor)                      hcir.vit_train        differentiable ViT forward
    positive_masking_transform(pos)                            hcir.transform        hcir_positive_masking
    model.forward_momentum(masked_pos)   (no grad)             hcir.vit_engine       inference engine
    F.normalize x 4                                            torch (B x 512 rows; autograd)
    TripletMarginLoss(margin, p=2, eps=1e-7)                   hcir.train_ops        hcir_triplet_margin_fwd / _bwd
    NTXentLoss(temperature)(pos, anchor)                       hcir.losses           hcir_ntxent_fwd / _bwd
    F.mse_loss(pos, masked_pos)                                hcir.train_ops        hcir_mse_fwd / _bwd
    total = contrastive + 0.5 triplet + 0.2 mse                (:735-742, ablation switches kept)
    scaler.scale(total).backward(); unscale_; clip_grad_norm_(1.0); scaler.step; scaler.update    (:745-749)

    positive_transform(pos)  (RandomRotation + GaussianBlur)   hcir.transform        hcir_positive_transform
    projection head in train mode (Linear/BN/ReLU/Linear/BN)   hcir.head_train       hcir_gemm_f16 / _tn, hcir_bn1d_*

The step's DEFAULT path is the reference's default path: `positive_transform` on (ablation "No_pos_transform" turns
it off, :684-685), hard negatives mined once in the epoch that ends the warm-up with k from the previous epoch's margin
violations and cached per batch (:633-654), "fixed_hard" and "randomly" as in the reference.  What stays torch: the
optimizer (torch.optim.Adam from utils.get_optimizer, :108), GradScaler and clip_grad_norm_ — no vendor GEMM is left
in the step.  The data loader, the epoch bookkeeping, logging and checkpointing of the reference's Trainer are outside
the hot path.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from .losses import NTXentLoss
from .momentum import update_momentum
from .neg_sampling import NegSamplerRandomly, NegSamplerStatic
from .train_ops import TripletMarginLoss, mse_loss
from .transform import PositiveMaskingTransform, PositiveTransform


class SHAMTrainStep:
    def __init__(self, model, optimizer, scaler=None, temperature: float = 0.5, momentum: float = 0.99,
                 warm_up_epochs: int = 0, ablation: str = "None",
                 positive_transform: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 mask_ratio_range=(0.1, 0.5)):
        self.model = model
        self.optimizer = optimizer
        self.scaler = scaler
        self.momentum = momentum
        self.warm_up_epochs = warm_up_epochs
        self.ablation = ablation
        self.criterion1 = NTXentLoss(temperature=temperature)                                   # :93
        if ablation == "fixed_margin":
            self.triplet_loss_stage1 = self.triplet_loss_stage2 = TripletMarginLoss(margin=0.7, p=2, eps=1e-7)
        else:
            self.triplet_loss_stage1 = TripletMarginLoss(margin=0.7, p=2, eps=1e-7)             # :96
            self.triplet_loss_stage2 = TripletMarginLoss(margin=0.5, p=2, eps=1e-7)             # :97
        self.positive_masking_transform = PositiveMaskingTransform(mask_ratio_range=mask_ratio_range)   # :99
        # HP/utils/transform.py:21-24 applied at :686 unless ablation == "No_pos_transform" (:684-685); a caller may
        # pass its own callable
        self.positive_transform = positive_transform if positive_transform is not None else PositiveTransform()
        self.negative_batch_idx = []
        self.total_k = 0

    def hard_negative_k(self, prev_margin_violations: float, batch_size: int) -> int:
        """:636-642 — x = max(2, round((1 - v) * 10)) with v = the previous epoch's margin violations per sample
        (Python's round: ties to even, as in the reference)."""
        v = prev_margin_violations / batch_size
        return max(2, round((1 - v) * 10))

    def _negatives(self, x_pos_1, epoch, batch_id, prev_margin_violations, negative_idx):
        """:627-680.  Stage 1 (epoch + 1 < warm_up_epochs) and the "randomly" ablation draw random negatives.  In the
        epoch that ends the warm-up (epoch + 1 == warm_up_epochs) every batch is mined ONCE with the momentum model
        (NegSamplerStatic, k from the schedule above, fixed at batch 0) and its indices are cached; every later epoch
        re-uses the cache by batch id (the reference's loader order is fixed).  "fixed_hard" takes the cached / mined
        indices in every epoch — before the mining epoch the cache is empty and the reference raises IndexError."""
        model = self.model
        if negative_idx is not None:                       # caller-supplied indices (tests, external caches)
            return x_pos_1[negative_idx]
        stage1 = self.warm_up_epochs > epoch + 1
        if self.ablation == "randomly" or (stage1 and self.ablation != "fixed_hard"):
            return NegSamplerRandomly(x_pos_1)                                                  # :631,660
        if (epoch + 1) == self.warm_up_epochs:
            if batch_id == 0:
                self.negative_batch_idx = []                                                    # :635
                self.total_k = self.hard_negative_k(prev_margin_violations, x_pos_1.shape[0])   # :637-642
            self.negative_batch_idx.append(NegSamplerStatic(model, x_pos_1, k=self.total_k))    # :644,670
        if batch_id >= len(self.negative_batch_idx):
            raise IndexError(f"no cached hard-negative indices for batch {batch_id}: they are mined in epoch "
                             f"{self.warm_up_epochs - 1} (epoch + 1 == warm_up_epochs), HP/src/pretrain_engine.py:633-654")
        return x_pos_1[self.negative_batch_idx[batch_id]]                                       # :654,680

    def __call__(self, batch: Dict[str, torch.Tensor], epoch: int = 0, negative_idx: Optional[torch.Tensor] = None,
                 generator=None, batch_id: int = 0, prev_margin_violations: float = 0.0) -> Dict[str, float]:
        """One optimisation step on {'anchor', 'pos1'} image batches [B,3,224,224] (HIP device): the body of the
        reference's batch loop; `batch_id` and `prev_margin_violations` are the loop's own variables (:618, :602)."""
        model, opt, scaler = self.model, self.optimizer, self.scaler
        model.train()
        opt.zero_grad()
        update_momentum(model.backbone, model.backbone_momentum, m=self.momentum)                # :621
        update_momentum(model.projection_head, model.projection_head_momentum, m=self.momentum)  # :622
        x_anchor, x_pos_1 = batch["anchor"], batch["pos1"]
        stage1 = self.warm_up_epochs > epoch + 1
        negative_samples = self._negatives(x_pos_1, epoch, batch_id, prev_margin_violations, negative_idx)

        pos_samples = x_pos_1 if self.ablation == "No_pos_transform" else self.positive_transform(x_pos_1)   # :684-687
        # :683,689,690 — model(neg), model(pos), model(anchor): one 3B-row backbone pass, the head once per view
        if hasattr(model, "forward_views"):
            neg_batch, pos_batch, anchor_batch = model.forward_views([negative_samples, pos_samples, x_anchor])
        else:
            neg_batch, pos_batch, anchor_batch = model(negative_samples), model(pos_samples), model(x_anchor)
        if self.ablation == "No masked positive":
            masked_pos_samples = pos_samples
        else:
            masked_pos_samples = self.positive_masking_transform(pos_samples, generator=generator)   # :694
        with torch.no_grad():
            masked_pos_batch = model.forward_momentum(masked_pos_samples)                       # :696

        neg_batch = F.normalize(neg_batch.float(), p=2, dim=1)                                  # :699-702
        pos_batch = F.normalize(pos_batch.float(), p=2, dim=1)
        anchor_batch = F.normalize(anchor_batch.float(), p=2, dim=1)
        masked_pos_batch = F.normalize(masked_pos_batch.float(), p=2, dim=1)
        with torch.no_grad():                                                                   # :703-714
            pos_dist = torch.norm(anchor_batch - pos_batch, p=2, dim=1)
            neg_dist = torch.norm(anchor_batch - neg_batch, p=2, dim=1)
            margin = self.triplet_loss_stage1.margin if stage1 else self.triplet_loss_stage2.margin
            violations = (pos_dist - neg_dist + margin > 0)

        out = {}
        triplet_loss = mse = None
        if self.ablation != "No_Triplet":                                                       # :718-723
            crit = self.triplet_loss_stage1 if stage1 else self.triplet_loss_stage2
            triplet_loss = crit(anchor_batch, pos_batch, neg_batch)
        contrastive_loss = self.criterion1(pos_batch, anchor_batch)                             # :725
        if self.ablation != "No_MSE":
            mse = mse_loss(pos_batch, masked_pos_batch, reduction="mean")                      # :730
        if self.ablation == "No_Triplet":                                                       # :735-742
            total_loss = contrastive_loss + 0.2 * mse
        elif self.ablation == "No_MSE":
            total_loss = contrastive_loss + 0.5 * triplet_loss
        else:
            total_loss = contrastive_loss + 0.5 * triplet_loss + 0.2 * mse

        if scaler is not None:                                                                  # :745-749
            scaler.scale(total_loss).backward()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            scaler.step(opt)
            scaler.update()
        else:
            total_loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            opt.step()
        zero = total_loss.new_zeros(())
        # one device-to-host read for the step's seven scalars
        vals = torch.stack([v.detach().float() for v in (
            total_loss, contrastive_loss, triplet_loss if triplet_loss is not None else zero,
            mse if mse is not None else zero, pos_dist.mean(), neg_dist.mean(), violations.sum())]).tolist()
        out.update(zip(("total", "contrastive", "triplet", "mse", "pos_dist", "neg_dist", "margin_violations"), vals))
        return out
