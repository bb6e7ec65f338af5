"""Deterministic synthetic weights shared by the golden generator, the tests and bench.py.

No trained checkpoint exists offline, and the reference's own
initialisation zeroes 109 of 304 tensors in the CIFAR config
(`AD/image_diffusion/nn.py:62-68`, `unet.py:310,389,705`), which would make every
parity test pass vacuously.  `synth_state_dict` therefore re-draws EVERY tensor
from a legacy `numpy.random.RandomState` stream (bit-stable across numpy
versions and machines) in state-dict key order:

  conv / linear weights  ~ U(-b, b),  b = gain / sqrt(fan_in)
  conv / linear biases   ~ U(-0.1, 0.1)
  GroupNorm weight       ~ 1 + 0.1 N(0,1);  GroupNorm bias ~ 0.1 N(0,1)

Data generation only (no compute path): used by bench.py, the tests and tools/make_goldens.py.
"""
from __future__ import annotations

import numpy as np
import torch


def synth_tensor(rs: np.random.RandomState, name: str, shape, gain: float = 1.0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    is_norm = (
        ".norm." in name
        or name.endswith("in_layers.0.weight")
        or name.endswith("in_layers.0.bias")
        or name.endswith("out_layers.0.weight")
        or name.endswith("out_layers.0.bias")
        or name in ("out.0.weight", "out.0.bias")
    )
    if len(shape) == 1:
        if is_norm and name.endswith("weight"):
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif is_norm:
            v = 0.1 * rs.standard_normal(shape)
        else:
            v = rs.uniform(-0.1, 0.1, size=shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        b = gain / np.sqrt(fan_in)
        v = rs.uniform(-b, b, size=shape)
    return torch.from_numpy(np.asarray(v, dtype=np.float32))


def synth_state_dict(shapes, seed: int, gain: float = 1.0):
    """shapes: ordered mapping name -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes.items():
        out[name] = synth_tensor(rs, name, shape, gain)
    return out


def randn(seed: int, *shape) -> torch.Tensor:
    """Seeded N(0,1) tensor from the legacy MT19937 stream (machine-independent)."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal(shape).astype(np.float32))


def rand_uniform(seed: int, lo: float, hi: float, *shape) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(lo, hi, size=shape).astype(np.float32))


def free_form_mask(seed: int, batch: int, height: int, width: int, coverage: float = 0.4, brush: int = 9) -> torch.Tensor:
    """Seeded random-walk brush mask [batch, 1, H, W] (bool, True = masked) with about `coverage` of the pixels masked: the
    "free-form mask" of BASELINE.json config 5 (the reference has no free-form mask generator; its masks are -2 sentinels at
    arbitrary positions, likelihoods.py:55-56, so any boolean mask is admissible input - SURVEY 8 config reconciliation)."""
    rs = np.random.RandomState(seed)
    out = np.zeros((batch, 1, height, width), dtype=bool)
    r = brush // 2
    target = coverage * height * width
    for b in range(batch):
        y, x = int(rs.randint(0, height)), int(rs.randint(0, width))
        ang = rs.uniform(0, 2 * np.pi)
        m = out[b, 0]
        while m.sum() < target:
            m[max(0, y - r): y + r + 1, max(0, x - r): x + r + 1] = True
            ang += rs.uniform(-0.6, 0.6)
            step = rs.uniform(2, 6)
            y = int(np.clip(y + step * np.sin(ang), 0, height - 1))
            x = int(np.clip(x + step * np.cos(ang), 0, width - 1))
            if rs.uniform() < 0.02:   # start a new stroke
                y, x = int(rs.randint(0, height)), int(rs.randint(0, width))
    return torch.from_numpy(out)
