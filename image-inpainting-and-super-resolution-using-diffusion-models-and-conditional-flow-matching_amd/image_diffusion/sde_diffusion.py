"""Counterpart of `AD/image_diffusion/sde_diffusion.py:101-244` (DDPM tables + `extract`).

The 14 tables are built once on the host exactly as the reference does (fp32 torch ops, so they are
bit-identical); the per-step arithmetic that the reference performs with them as 8-10 eager
elementwise kernels lives in csrc/steps.hip and is driven by `image_diffusion.sampling`.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn.functional as F
from torch import nn

Network = Callable[[torch.Tensor, torch.Tensor], torch.Tensor]

bm = 0.1
bd = 20

TABLE_NAMES = (
    "alphas", "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "recip_sqrt_m1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def int_b(t):
    return bm * t + (bd - bm) * t ** 2 / 2


def beta(t):
    return bm + (bd - bm) * t


def extract(a, t, x_shape):
    """sde_diffusion.py:101-104."""
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


class DDPM(nn.Module):
    """sde_diffusion.py:107-167.  `DDPM(Ns <= 20)` has non-finite entries exactly like the reference
    (betas[-1] = 20/Ns >= 1): this is reproduced, not repaired (SURVEY.md finding 4)."""

    def __init__(self, Ns: int):
        super().__init__()
        self.Ns = Ns
        self.tmin = 0.00001
        self.tmax = 1.0
        self.ts = torch.linspace(self.tmin, self.tmax, Ns, dtype=torch.float32)
        reg = lambda name, val: self.register_buffer(name, val.to(torch.float32))
        betas = beta(self.ts) / Ns
        reg("alphas", 1.0 - betas)
        alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.0)
        reg("betas", betas)
        reg("alphas_cumprod", alphas_cumprod)
        reg("alphas_cumprod_prev", alphas_cumprod_prev)
        reg("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod))
        reg("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - alphas_cumprod))
        reg("log_one_minus_alphas_cumprod", torch.log(1.0 - alphas_cumprod))
        reg("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / alphas_cumprod))
        reg("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / alphas_cumprod - 1))
        reg("recip_sqrt_m1_alphas_cumprod", 1.0 / torch.sqrt(1 - alphas_cumprod))
        posterior_variance = betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod)
        reg("posterior_variance", posterior_variance)
        reg("posterior_log_variance_clipped", torch.log(posterior_variance.clamp(min=1e-20)))
        reg("posterior_mean_coef1", betas * torch.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod))
        reg("posterior_mean_coef2", (1.0 - alphas_cumprod_prev) * torch.sqrt(self.alphas) / (1.0 - alphas_cumprod))
        self._host = None

    def host_tables(self):
        """CPU fp32 copies of the buffers (per-step scalars are passed to the kernels by value)."""
        if self._host is None:
            self._host = {n: getattr(self, n).detach().to("cpu", torch.float32).contiguous() for n in TABLE_NAMES}
        return self._host

    def _apply(self, fn, *a, **k):
        self._host = None
        return super()._apply(fn, *a, **k)
