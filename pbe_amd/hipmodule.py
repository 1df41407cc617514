"""Base class for the host-side mirror modules: fp32 master parameters under the reference's
names (so ``load_state_dict`` of a Paint-by-Example checkpoint works unchanged) plus a lazily
built cache of kernel-ready fp16 packs that is dropped whenever the parameters may have changed
(``load_state_dict``, ``.to()``, ``.cuda()``, ``.half()`` ...)."""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn


class HipModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.__dict__["_pk_cache"] = None
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate_packs())

    def invalidate_packs(self):
        for m in self.modules():
            if isinstance(m, HipModule):
                m.__dict__["_pk_cache"] = None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.__dict__["_pk_cache"] = None
        return r

    def _pack(self) -> SimpleNamespace:          # overridden: build fp16 / fp32 kernel operands
        return SimpleNamespace()

    def pk(self) -> SimpleNamespace:
        c = self.__dict__.get("_pk_cache")
        if c is None:
            with torch.no_grad():
                c = self._pack()
            self.__dict__["_pk_cache"] = c
        return c


def f32(p: torch.Tensor) -> torch.Tensor:
    return p.detach().float().contiguous()


def require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        from .lib import PbeError
        raise PbeError(f"{what}: the HIP path needs tensors on an MI355X (no CPU fallback); got device {t.device}")
