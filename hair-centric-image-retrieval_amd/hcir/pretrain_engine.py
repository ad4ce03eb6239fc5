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

What stays torch: the optimizer (torch.optim.Adam from utils.get_optimizer, :108), GradScaler, clip_grad_norm_ and
the lightly projection head (nn.Linear / BatchNorm1d modules, as in the reference).  The data loader, the epoch
bookkeeping, logging and checkpointing of the reference's Trainer are outside the hot path.
`positive_transform` (RandomRotation + GaussianBlur augmentation, HP/utils/transform.py:21-24) is an augmentation
policy, not arithmetic of the step: the caller passes its own callable (default: identity = the reference's
"No_pos_transform" ablation).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from .losses import NTXentLoss
from .momentum import update_momentum
from .neg_sampling import NegSamplerRandomly, NegSamplerStatic
from .train_ops import TripletMarginLoss, mse_loss
from .transform import PositiveMaskingTransform


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
        self.positive_transform = positive_transform
        self.negative_batch_idx = []

    def __call__(self, batch: Dict[str, torch.Tensor], epoch: int = 0, negative_idx: Optional[torch.Tensor] = None,
                 generator=None) -> Dict[str, float]:
        """One optimisation step on {'anchor', 'pos1'} image batches [B,3,224,224] (HIP device).  Stage 1
        (epoch + 1 < warm_up_epochs) draws random negatives; otherwise `negative_idx` (NegSamplerStatic's output for
        this batch, :643,652) selects them, or they are mined here with k = 2."""
        model, opt, scaler = self.model, self.optimizer, self.scaler
        model.train()
        opt.zero_grad()
        update_momentum(model.backbone, model.backbone_momentum, m=self.momentum)                # :621
        update_momentum(model.projection_head, model.projection_head_momentum, m=self.momentum)  # :622
        x_anchor, x_pos_1 = batch["anchor"], batch["pos1"]
        stage1 = self.warm_up_epochs > epoch + 1
        if stage1 or self.ablation == "randomly":
            negative_samples = NegSamplerRandomly(x_pos_1)                                      # :631,660
        else:
            if negative_idx is None:
                negative_idx = NegSamplerStatic(model, x_pos_1, k=2)                            # :646 (k from :638-642)
            negative_samples = x_pos_1[negative_idx]                                            # :654

        neg_batch = model(negative_samples)                                                     # :683
        pos_samples = x_pos_1 if self.positive_transform is None else self.positive_transform(x_pos_1)   # :684-687
        pos_batch = model(pos_samples)                                                          # :689
        anchor_batch = model(x_anchor)                                                          # :690
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
        out.update(total=float(total_loss), contrastive=float(contrastive_loss),
                   triplet=float(triplet_loss) if triplet_loss is not None else 0.0,
                   mse=float(mse) if mse is not None else 0.0, pos_dist=float(pos_dist.mean()),
                   neg_dist=float(neg_dist.mean()), margin_violations=float(violations.sum()))
        return out
