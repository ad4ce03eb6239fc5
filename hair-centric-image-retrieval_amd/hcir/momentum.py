"""hcir.momentum — update_momentum (lightly.models.utils.update_momentum as the reference calls it,
HP/src/pretrain_engine.py:618-619; HP/utils/utils.py:113-115) as one HIP launch over all parameters.

    update_momentum(model, model_ema, m):   ema = ema * m + p * (1 - m)     for every parameter pair

The chunk table (device pointers of 64 Ki-element pieces of every tensor pair) is built once per
(model, model_ema) pair and rebuilt when a parameter's storage moves.  fp32 parameters on a HIP device
only; results are bit-identical to the reference expression.  No CPU path.
"""
from __future__ import annotations

import weakref

import numpy as np
import torch

from . import _lib
from ._lib import HcirError, check

_CHUNK = 65536
_tables: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _build(params, params_ema, device):
    dst, src, cnt = [], [], []
    for p, e in zip(params, params_ema):
        if p.shape != e.shape:
            raise HcirError("update_momentum: parameter shapes differ")
        if p.dtype != torch.float32 or e.dtype != torch.float32:
            raise HcirError("update_momentum: fp32 parameters only")
        if not (p.is_cuda and e.is_cuda):
            raise HcirError(f"update_momentum: parameters are on {p.device}/{e.device}; HIP device only")
        if not (p.is_contiguous() and e.is_contiguous()):
            raise HcirError("update_momentum: parameters must be contiguous")
        n = p.numel()
        for off in range(0, n, _CHUNK):
            dst.append(e.data_ptr() + 4 * off)
            src.append(p.data_ptr() + 4 * off)
            cnt.append(min(_CHUNK, n - off))
    if not dst:
        return None
    tab = torch.from_numpy(np.array([dst, src, cnt], dtype=np.int64)).to(device)
    return tab


def update_momentum(model: torch.nn.Module, model_ema: torch.nn.Module, m: float) -> None:
    params = [p for p in model.parameters()]
    params_ema = [p for p in model_ema.parameters()]
    if len(params) != len(params_ema):
        raise HcirError("update_momentum: models have different parameter lists")
    if not params:
        return
    device = params_ema[0].device
    key = tuple((p.data_ptr(), e.data_ptr(), p.numel()) for p, e in zip(params, params_ema))
    cached = _tables.get(model_ema)
    if cached is None or cached[0] != key:
        cached = (key, _build(params, params_ema, device))
        _tables[model_ema] = cached
    tab = cached[1]
    if tab is None:
        return
    # (1.0 - m) in Python floats, then both scalars rounded to fp32: what torch does for `tensor * python_float`
    mm, om = float(np.float32(m)), float(np.float32(1.0 - m))
    st = torch.cuda.current_stream(device).cuda_stream
    check(_lib.lib().hcir_ema_update(tab[0].data_ptr(), tab[1].data_ptr(), tab[2].data_ptr(), tab.shape[1], mm, om, st),
          "hcir_ema_update")
    # the kernel wrote through raw pointers: bump the version counters so that caches keyed on them
    # (hcir.vit_engine.EngineCache) see the new weights
    for e in params_ema:
        torch.autograd.graph.increment_version(e)
