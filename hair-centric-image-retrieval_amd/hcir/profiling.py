"""Per-op HIP-event timing on the stream the kernels are launched on."""
from __future__ import annotations

from collections import OrderedDict

import torch


class EventProfiler:
    """mark(name) attributes the time since the previous mark to `name` (HIP events on
    torch's current stream, which is the stream hcir ops launch on)."""

    def __init__(self):
        self._spans = []
        self._prev = None

    def start(self) -> None:
        self._prev = torch.cuda.Event(enable_timing=True)
        self._prev.record()

    def mark(self, name: str) -> None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if self._prev is not None:
            self._spans.append((name, self._prev, e))
        self._prev = e

    def summary(self) -> "OrderedDict[str, dict]":
        torch.cuda.synchronize()
        out: "OrderedDict[str, dict]" = OrderedDict()
        for name, a, b in self._spans:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0})
            d["calls"] += 1
            d["ms"] += a.elapsed_time(b)
        return out
