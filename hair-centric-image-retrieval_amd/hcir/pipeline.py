"""hcir.pipeline — several query batches in flight on separate HIP streams.

At the reference's query-batch sizes (64 images: SURVEY §8d C2; 256: HP/scripts/classification/celebA/knn_our_vit.sh)
one batch does not fill 256 CUs: proj / fc2 are 150 tiles of a round, every launch waits ~5 us for its predecessor
(the dependent-dispatch gap, 17 % of a 64-image step), and each GEMM ends in a partial round of tiles.  The batches
of a retrieval run are independent, so two of them run side by side: each on its own HIP stream with its own engine
buffers (`VitEngine` slot), the embed + scan of one filling the gaps and tails of the other.  Results come back in
submission order; every batch is finished (certified, merged) before it is returned.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch


class StreamPipeline:
    """submit(x) enqueues embed + top-k of one query batch on the next stream and returns the result of the batch
    submitted `depth` calls earlier (None while the pipeline fills); drain() returns what is still in flight."""

    def __init__(self, backbone, gallery, k: int, depth: int = 2, device=None):
        self.backbone, self.gallery, self.k = backbone, gallery, int(k)
        self.device = torch.device(device if device is not None else "cuda")
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self.pending: List[Optional[Callable]] = [None] * depth
        self.turn = 0

    def _finish(self, slot: int):
        h = self.pending[slot]
        if h is None:
            return None
        self.pending[slot] = None
        with torch.cuda.stream(self.streams[slot]):
            return h.finish() if hasattr(h, "finish") else self.gallery.search_finish(h)

    def submit(self, x: torch.Tensor):
        slot = self.turn
        self.turn = (self.turn + 1) % len(self.streams)
        out = self._finish(slot)  # the batch this slot held: its buffers are free again after this
        s = self.streams[slot]
        s.wait_stream(torch.cuda.current_stream(self.device))  # x was produced on the caller's stream
        with torch.cuda.stream(s), torch.no_grad():
            e32, e16 = self.backbone.forward_cls(x, l2_normalize=True, want_f16=True, slot=slot)
            x.record_stream(s)
            self.pending[slot] = self.gallery.search_begin(e32, self.k, q16=e16)
        return out

    def drain(self):
        outs = []
        for i in range(len(self.streams)):
            slot = (self.turn + i) % len(self.streams)
            r = self._finish(slot)
            if r is not None:
                outs.append(r)
        return outs
