"""Counterpart of `AD/image_diffusion/nn.py` for the sampling path.

`timestep_embedding` runs the HIP kernel (csrc/embed.hip); GroupNorm32 / SiLU / conv / linear are not
separate modules here: they are fused inside the U-Net plan (csrc/unet_engine.hip).  `normalization`
is offered as a functional HIP op for parity tests.
"""
from __future__ import annotations

import torch

from mi355.ops import default_ops


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """nn.py:97-115 - sinusoidal embeddings [N, dim] of (possibly fractional) timesteps, fp32."""
    return default_ops.timestep_embedding(timesteps.float().contiguous(), dim, float(max_period))


def group_norm32(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, silu: bool = False) -> torch.Tensor:
    """GroupNorm32(32, C) (nn.py:11-13,87-94), optionally fused with SiLU; x is [N, C, ...] fp32."""
    return default_ops.groupnorm(x.float().contiguous(), weight, bias, 32, 1e-5, silu)


def zero_module(module):
    """nn.py:62-68."""
    for p in module.parameters():
        p.detach().zero_()
    return module


def update_ema(target_params, source_params, rate=0.99):
    """nn.py:49-59 (host-side parameter plumbing)."""
    for targ, src in zip(target_params, source_params):
        targ.detach().mul_(rate).add_(src, alpha=1 - rate)
