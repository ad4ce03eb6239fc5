"""oracle.metrics — numpy / pure-Python restatement of the reference's retrieval metrics.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows experiments/DualViewHair/scripts/quantitative_eval.py:194-209 (per-query loop) and :228-234
(means): the same Python-float arithmetic in the same order, with integer ids in place of the
reference's path strings.
"""
from __future__ import annotations

from typing import Dict, List, Sequence


def evaluate_ids(retrieved: Sequence[Sequence[int]], gt_lists: Sequence[Sequence[int]], Ks=(10, 20, 50)):
    recall_at_k = {k: 0 for k in Ks}
    ap_at_k: Dict[int, List[float]] = {k: [] for k in Ks}
    hit_rows = {k: [] for k in Ks}
    total_queries = 0
    for ret, gt_list in zip(retrieved, gt_lists):
        ret = [int(x) for x in ret]
        gt_list = [int(x) for x in gt_list]
        for k in Ks:
            top_k_preds = ret[:k]                                     # :195
            hit = any(gt in top_k_preds for gt in gt_list)            # :198
            if hit:
                recall_at_k[k] += 1
            hit_rows[k].append(1 if hit else 0)
            hits, sum_precisions = 0, 0                               # :202-207
            for i, p in enumerate(top_k_preds):
                if p in gt_list:
                    hits += 1
                    sum_precisions += hits / (i + 1)
            ap = sum_precisions / min(len(gt_list), k) if gt_list else 0.0   # :208
            ap_at_k[k].append(ap)
        total_queries += 1
    return {
        "mAP": {k: sum(ap_at_k[k]) / len(ap_at_k[k]) if ap_at_k[k] else 0 for k in Ks},          # :229
        "Recall": {k: recall_at_k[k] / total_queries if total_queries > 0 else 0 for k in Ks},  # :230
        "total_queries": total_queries,
        "ap": ap_at_k, "hit": hit_rows,
    }
