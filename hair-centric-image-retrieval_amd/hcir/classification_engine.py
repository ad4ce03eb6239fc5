"""hcir.classification_engine — Classifier.knn_eval on the MI355X hot path
(HP/src/classification_engine.py:18-98).

Differences from the reference, all behind the same interface:
  * embeddings stay on the device (no per-batch .cpu(), :51,63); F.normalize is hcir_l2_normalize;
  * the brute-force cosine kNN is ONE hcir_sim_topk call for max(ks) neighbours instead of a
    sklearn fit + full distance matrix per k (:79-82 recomputes it 7 times); every k of the
    sweep is a prefix of that list;
  * the uniform-weight vote (mode of the k neighbour labels, smallest label on ties — sklearn
    KNeighborsClassifier.predict) of ALL k of the sweep is one hcir_knn_vote launch over that list, and
    accuracy / confusion-matrix counts come from hcir_confusion_matrix: labels and predictions stay in HBM;
    only the per-k report (text) is assembled on the host.
linear_probe_eval / save_umap / compute_intra_inter_variance are sklearn/umap analytics outside
the hot path and raise NotImplementedError.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops


def knn_vote(nbr_labels: np.ndarray, nclass: int) -> np.ndarray:
    """mode over axis 1, smallest label wins ties (scipy.stats.mode as sklearn uses it)."""
    counts = np.zeros((nbr_labels.shape[0], nclass), dtype=np.int64)
    rows = np.repeat(np.arange(nbr_labels.shape[0]), nbr_labels.shape[1])
    np.add.at(counts, (rows, nbr_labels.reshape(-1)), 1)
    return counts.argmax(axis=1)


def _report(y_true, y_pred):
    try:
        from sklearn.metrics import classification_report, confusion_matrix
        return classification_report(y_true, y_pred), confusion_matrix(y_true, y_pred)
    except ImportError:  # reporting only; not part of the retrieval path
        labels = np.unique(np.concatenate([y_true, y_pred]))
        cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
        lut = {l: i for i, l in enumerate(labels)}
        for t, p in zip(y_true, y_pred):
            cm[lut[t], lut[p]] += 1
        return "(scikit-learn not installed: classification_report unavailable)", cm


class Classifier:
    def __init__(self, model, train_loader, test_loader, args):
        self.model = model.to(args.device)
        self.train_loader = train_loader
        self.test_loader = test_loader
        self.device = args.device
        self.mode = args.mode
        self.mode_model = args.model
        if args.mode == "SHAM":
            self.save_path = os.path.join(args.save_path, f"{self.mode}_{self.mode_model}_{args.SHAM_mode}")
        else:
            self.save_path = os.path.join(args.save_path, f"{self.mode}_{self.mode_model}")
        os.makedirs(self.save_path, exist_ok=True)
        self.training_features = []
        self.training_labels = []
        self.testing_features = []
        self.testing_labels = []

    def _embed(self, loader, feats, labels):
        for batch in loader:
            images, lab = batch[0], batch[1]
            if hasattr(images, "decode"):  # hcir.dataloader.EncodedBatch: compressed files, decoded on the device
                images = images.decode(self.device)
            images = images.to(self.device)
            if images.dtype == torch.uint8:  # raw RGB8 windows [B,H,W,3]: ToTensor + Normalize on the device
                from .transform import knn_transform_u8
                images = knn_transform_u8(images)
            f = self.model.extract_features(images)
            feats.append(ops.l2_normalize(f.float().contiguous()))  # F.normalize(dim=1), stays in HBM
            labels.append(torch.as_tensor(lab))

    def extracting_features(self):
        self.model.eval()
        # (re)start from empty lists so that a second evaluation on the same object works
        self.training_features, self.training_labels = [], []
        self.testing_features, self.testing_labels = [], []
        with torch.no_grad():
            self._embed(self.train_loader, self.training_features, self.training_labels)
            self._embed(self.test_loader, self.testing_features, self.testing_labels)
        self.training_features = torch.cat(self.training_features)
        self.training_labels = torch.cat(self.training_labels)
        self.testing_features = torch.cat(self.testing_features)
        self.testing_labels = torch.cat(self.testing_labels)

    def kneighbors(self, k):
        """(cosine distance, index) of the k nearest training rows of every test row —
        KNeighborsClassifier(metric='cosine').kneighbors: both sides re-normalised,
        d = clip(1 - S, 0, 2)."""
        g, q = self.training_features, self.testing_features
        val, idx = ops.sim_topk(q, g, k, q_inv_norm=ops.row_invnorm(q, 1e-30),
                                g_inv_norm=ops.row_invnorm(g, 1e-30))
        return (1.0 - val).clamp_(0.0, 2.0), idx

    def knn_eval(self, ks=(5, 10, 20, 27, 30, 40, 642)):
        print(f"Evaluating on KNN classifier with {self.device}")
        self.extracting_features()
        file_path = os.path.join(self.save_path, "knn_evaluation_results.txt")
        with open(file_path, "w") as f:
            f.write("KNN Evaluation Results\n")
            f.write("=" * 50 + "\n\n")
        n_train = self.training_features.shape[0]
        valid = sorted({k for k in ks if k <= n_train})
        y_true = self.testing_labels.numpy()
        nclass = int(max(self.training_labels.max(), self.testing_labels.max())) + 1
        preds = {}
        if valid:
            from . import metrics
            _, idx = self.kneighbors(valid[-1])
            dev = idx.device
            pred = metrics.knn_vote(idx, self.training_labels.to(dev), valid, nclass)   # [len(valid), n_test]
            yt = self.testing_labels.to(dev)
            for j, k in enumerate(valid):
                cm = metrics.confusion_matrix(yt, pred[j].contiguous(), nclass)
                preds[k] = (pred[j].cpu().numpy(), float(cm.diagonal().sum().item()) / max(len(y_true), 1))
        for k in ks:
            if k > n_train:  # sklearn raises here and the reference's sweep stops
                raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, "
                                 f"n_samples_fit = {n_train}, n_samples = {len(y_true)}")
            y_pred, acc = preds[k]
            report, cm = _report(y_true, y_pred)
            with open(file_path, "a") as f:
                f.write(f"Results for k={k}\n")
                f.write("-" * 40 + "\n")
                f.write(f"Accuracy: {acc:.4f}\n\n")
                f.write("Classification Report:\n")
                f.write(report + "\n\n")
                f.write("Confusion Matrix:\n")
                f.write(np.array2string(cm) + "\n\n")
                f.write("=" * 50 + "\n\n")
            print(f"Appended results for k={k}")
        print(f"\nAll results saved in: {file_path}")

    def linear_probe_eval(self, *a, **k):
        raise NotImplementedError("linear probe is sklearn LogisticRegression analytics, outside the hot path")

    def save_umap(self, *a, **k):
        raise NotImplementedError("UMAP visualisation is outside the hot path")

    def compute_intra_inter_variance(self, *a, **k):
        raise NotImplementedError("intra/inter-class variance analytics are outside the hot path")
