#!/usr/bin/env python3
"""knn_classification.py — the reference's kNN evaluation CLI (HP/knn_classification.py) on the
MI355X hot path.  Same flags (HP/knn_classification.py:47-67) and the same output file
`<save_path>/<mode>_<model>[_<SHAM_mode>]/knn_evaluation_results.txt`.

Modes on the hot path: SHAM (resnet18 / resnet50 / vit_b_16), simclr, mae.  The other --mode
values name SSL baselines that are out of scope: accepted by argparse,
rejected with a clear error.  --eval_type knn is built; the other eval types are host analytics.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import random

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Self-supervised/Supervised Trainer Arguments")
    parser.add_argument('--save_path', type=str, default='classification_output_dir', help='Path to save model checkpoint')
    parser.add_argument('--size', type=int, default=224, help="Image size for training")
    parser.add_argument('--train_annotation', type=str, help='Path to training annotation file')
    parser.add_argument('--test_annotation', type=str, help='Path to testing annotation file')
    parser.add_argument('--img_dir', type=str, help='Path to image directory')
    parser.add_argument('--batch_size', type=int, default=32, help='Batch size')
    parser.add_argument('--mode', type=str, default='simclr_supcon',
                        choices=['mae', 'simclr', 'simclr_supcon', 'dinov2', 'simMIM', 'siaMIM', "SHAM", "DenseCL", "MSN"])
    parser.add_argument('--model', type=str, default='resnet18', choices=['resnet18', 'resnet50', "vit_b_16"])
    parser.add_argument('--checkpoint_path', type=str, default=None)
    parser.add_argument('--device', type=str, default='cuda', help='Device to use: cuda or cpu')
    parser.add_argument('--SHAM_mode', type=str, default="embedding", choices=['embedding', 'reconstruction'])
    parser.add_argument('--seed', type=int, default=42, help='Random seed')
    parser.add_argument('--config', type=str, help='Optional path to YAML config file (overrides args)')
    parser.add_argument('--num_workers', type=int, default=4)
    parser.add_argument('--eval_type', default=None, type=str,
                        choices=["knn", "linear_prob", "visualization", "inter_intra_distance"])
    return parser.parse_args(argv)


def merge_config_with_args(args):
    if args.config and os.path.exists(args.config):
        with open(args.config, 'r') as f:
            config_file = yaml.safe_load(f)
        for key, value in config_file.items():
            if getattr(args, key, None) is None:
                setattr(args, key, value)
    return args


def set_seed(seed):
    """HP/utils/utils.py:105-111."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def build_model(args):
    from hcir.backbone import MAE, SimCLR, vit_base_patch16_224
    from hcir.main_backbone import SHAM2

    def load(path):
        return torch.load(path, map_location="cpu", weights_only=False)

    if args.mode == "simclr":
        model = SimCLR(model=args.model)
        if args.checkpoint_path:
            model.load_state_dict(load(args.checkpoint_path)['model_state_dict'])
    elif args.mode == "SHAM":
        model = SHAM2(model=args.model)
        if args.checkpoint_path:
            sd = load(args.checkpoint_path)
            # ViT checkpoints are trainer dicts, ResNet ones raw state_dicts (HP/knn_classification.py:139-142)
            model.load_state_dict(sd['model_state_dict'] if args.model == "vit_b_16" else sd)
    elif args.mode == "mae":
        model = MAE(vit_base_patch16_224())
        if args.checkpoint_path:
            model.load_state_dict(load(args.checkpoint_path)['model_state_dict'], strict=False)  # decoder not built
    else:
        raise SystemExit(f"--mode {args.mode}: this SSL baseline is outside the MI355X hot path "
                         "(SURVEY.md §2.1 row 4); built modes: SHAM, simclr, mae")
    if args.checkpoint_path:
        print("Model weights loaded!")
    return model


def main(args):
    from hcir.classification_engine import Classifier
    from hcir.dataloader import CustomDataset
    # HCIR_HOST_TRANSFORM=1: the reference's host-side knn_transform (fp32 tensors from the workers).
    # Default: the workers ship the RGB8 centre window and ToTensor + Normalize run on the device
    # (hcir_knn_transform_u8, bit-identical arithmetic, 4x less H2D traffic)
    if os.environ.get("HCIR_HOST_TRANSFORM", "0") == "1":
        from hcir.transform import knn_transform
    else:
        from hcir.transform import center_window_u8 as knn_transform

    if os.environ.get("HCIR_DEVICE_DECODE", "1") == "1" and os.environ.get("HCIR_HOST_TRANSFORM", "0") != "1":
        # default: the workers read + stage the files, JPEG decode / CenterCrop / ToTensor / Normalize run on the device
        from hcir.dataloader import EncodedDataset, collate_encoded
        train_dataset = EncodedDataset(args.train_annotation, args.img_dir)
        test_dataset = EncodedDataset(args.test_annotation, args.img_dir)
        kw = dict(collate_fn=collate_encoded)
    else:
        train_dataset = CustomDataset(args.train_annotation, args.img_dir, knn_transform)
        test_dataset = CustomDataset(args.test_annotation, args.img_dir, knn_transform)
        kw = {}
    train_loader = DataLoader(train_dataset, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers, **kw)
    test_loader = DataLoader(test_dataset, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers, **kw)
    model = build_model(args)
    trainer = Classifier(model, train_loader, test_loader, args)
    if args.eval_type == "knn":
        trainer.knn_eval()
    elif args.eval_type == "linear_prob":
        trainer.linear_probe_eval()
    elif args.eval_type == "visualization":
        trainer.save_umap(split="test")
    elif args.eval_type == "inter_intra_distance":
        trainer.compute_intra_inter_variance(split="both")


if __name__ == "__main__":
    args = parse_args()
    args = merge_config_with_args(args)
    set_seed(args.seed)
    main(args)
