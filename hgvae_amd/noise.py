"""Where dropout masks and normal draws come from.

Default: nothing is materialised - kernels derive every keep decision / normal draw
from Philox4x32 (7 rounds for dropout bits, 10 for normal draws) keyed by ``(seed, offset)``; the offset advances once per stochastic
layer call so that forward and backward regenerate identical noise.

For parity tests an explicit :class:`NoiseTape` replays pre-drawn tensors in call order
(the reference draws them with ``torch.nn.Dropout`` / ``torch.randn_like``,
gvae/gvae_skipcon.py:62,140).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional

import torch


STEP_STRIDE = 1 << 40   # device-counter increment per step; host offsets stay below it


class PhiloxNoise:
    """Counter-based noise: every stochastic layer call gets its own ``offset``.

    ``device_counter`` (optional, int64[1] on the GPU) is added to the offset inside the kernels.
    A captured graph replays fixed host offsets, so the training step advances this counter on the
    device (``advance()``) to draw fresh noise on every replay.
    """

    def __init__(self, seed: Optional[int] = None, device_counter: Optional[torch.Tensor] = None):
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 64) - 1)
        self.offset = 0
        self.device_counter = device_counter

    def _next(self) -> int:
        # 2**20 counter values per call keep the (offset + k) sub-streams of one call disjoint
        self.offset = (self.offset + (1 << 20)) % STEP_STRIDE
        return self.offset

    def _ctr(self):
        return None if self.device_counter is None else self.device_counter.data_ptr()

    def dropout(self, shape, device):
        return None, self.seed, self._next(), self._ctr()

    def normal(self, shape, device):
        return None, self.seed, self._next(), self._ctr()

    def advance(self) -> None:
        """Move the device counter to the next step (an in-place device op: capturable)."""
        if self.device_counter is not None:
            self.device_counter.add_(STEP_STRIDE)


class NoiseTape:
    """Explicit noise, consumed in call order: {0,1} masks for dropout, normal draws for eps."""

    def __init__(self, items: List[torch.Tensor]):
        self.items = list(items)
        self.pos = 0

    def _pop(self, shape, device):
        if self.pos >= len(self.items):
            raise RuntimeError("NoiseTape exhausted")
        t = self.items[self.pos]
        self.pos += 1
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"NoiseTape item {self.pos - 1} has shape {tuple(t.shape)}, layer expects {tuple(shape)}")
        return t.to(device)

    def dropout(self, shape, device):
        return self._pop(shape, device).to(torch.uint8).contiguous(), 0, 0, None

    def normal(self, shape, device):
        return self._pop(shape, device).to(torch.float32).contiguous(), 0, 0, None


DEFAULT_SEED = 0x5EED
_default = PhiloxNoise(seed=DEFAULT_SEED)
_stack: list = []


def manual_seed(seed: int) -> None:
    """Reset the default Philox stream."""
    global _default
    _default = PhiloxNoise(seed)


def current_noise():
    return _stack[-1] if _stack else _default


@contextlib.contextmanager
def use_noise(source):
    _stack.append(source)
    try:
        yield source
    finally:
        _stack.pop()
