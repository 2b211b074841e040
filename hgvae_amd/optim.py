"""Adam (and gradient-norm clipping) as one launch over flat parameter / gradient buffers.

The reference builds ``torch.optim.Adam(model.parameters(), lr=...)`` (gvae/gvae_skipcon.py:294,
autoencoder/vae_with_pool.py:270) and, in the custom-pool loop, clips the gradient norm first
(autoencoder/vae_custom_pool.py:228-229).  These models have ~40 small parameter tensors
(0.2-0.55 M floats in total), so the update is launch-bound; here every parameter is a view of one
contiguous buffer, every gradient a view of the :class:`FlatGradBucket` buffer, and the update is a
single kernel with torch's Adam arithmetic.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import _lib
from .distributed import FlatGradBucket


class FlatAdam:
    """Drop-in for ``torch.optim.Adam`` (no amsgrad / maximize) on the HIP update kernel.

    ``max_grad_norm`` fuses ``clip_grad_norm_(parameters, max_grad_norm)`` into the step.
    ``capturable=True`` keeps the step count on the device so that the step can sit inside a
    captured HIP graph.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, bucket: Optional[FlatGradBucket] = None,
                 max_grad_norm: Optional[float] = None, capturable: bool = False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        _lib.require_cuda(*self.params)
        if any(p.dtype != torch.float32 for p in self.params):
            raise _lib.HgvError("FlatAdam handles float32 parameters only")
        self.bucket = bucket if bucket is not None else FlatGradBucket(self.params)
        if [id(p) for p in self.bucket.params] != [id(p) for p in self.params]:
            raise ValueError("the gradient bucket must hold the same parameters in the same order")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:                      # every parameter becomes a view of the flat buffer
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev) if capturable else None
        self.max_grad_norm = max_grad_norm
        self._clip = torch.zeros(2, dtype=torch.float32, device=dev)      # [norm, scale]
        self._ws = torch.empty(_lib.load().hgv_grad_norm_ws(), dtype=torch.uint8, device=dev)
        self.param_groups = [{"params": self.params, "lr": lr, "betas": tuple(betas), "eps": eps,
                              "weight_decay": weight_decay, "capturable": capturable}]

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.bucket.zero()

    @property
    def last_grad_norm(self) -> torch.Tensor:
        """Gradient norm measured by the last clipped step (device scalar)."""
        return self._clip[0]

    def step(self) -> None:
        g = self.param_groups[0]
        self.bucket.gather()
        n = self.flat.numel()
        st = _lib.stream()
        scale_ptr = None
        if self.max_grad_norm is not None:
            _lib.call("hgv_grad_norm_clip", self.bucket.flat.data_ptr(), n, float(self.max_grad_norm),
                      self._clip.data_ptr(), self._clip.data_ptr() + 4, self._ws.data_ptr(), self._ws.numel(), st)
            scale_ptr = self._clip.data_ptr() + 4
        self.step_count += 1
        if self.step_dev is not None:
            self.step_dev.add_(1)
        _lib.call("hgv_adam_step", self.flat.data_ptr(), self.bucket.flat.data_ptr(), self.exp_avg.data_ptr(),
                  self.exp_avg_sq.data_ptr(), n, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                  float(g["eps"]), float(g["weight_decay"]), self.step_count, _lib.ptr(self.step_dev), scale_ptr, st)
        # the kernel wrote through raw pointers: tell autograd-side caches (packed weights) that the values changed
        torch.autograd.graph.increment_version(self.params)

    def state_dict(self) -> dict:
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, state: dict) -> None:
        self.step_count = int(state["step"])
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        if self.step_dev is not None:
            self.step_dev.fill_(self.step_count)
        self.param_groups[0].update(state["param_groups"][0])
