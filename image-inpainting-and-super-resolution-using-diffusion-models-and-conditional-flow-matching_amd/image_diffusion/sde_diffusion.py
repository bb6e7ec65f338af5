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

    # ---- per-sample step arithmetic (sde_diffusion.py:214-244): `i` is a long [B] index tensor; the coefficient gather is
    # `extract`, the multiply-adds are ONE fused HIP kernel (mi355_lincomb_per_sample) instead of 3-5 eager ones.  The samplers in
    # image_diffusion.sampling use the scalar-coefficient step kernels; these methods serve callers that build their own
    # x0_model closure (trainer2.py / loss_functions.py style). ----
    @staticmethod
    def _coef(table, i, x):
        return extract(table, i.to(table.device), x.shape).reshape(-1).to(x.device, torch.float32).contiguous()

    @staticmethod
    def _lin(x, a, y=None, b=None):
        from mi355.ops import default_ops

        xs = x.float().contiguous()
        return default_ops.lincomb_per_sample(xs, a, None if y is None else y.float().contiguous(), b)

    def score_from_x0(self, x_0, i):
        """sde_diffusion.py:214-217."""
        return self._lin(x_0, -self._coef(self.recip_sqrt_m1_alphas_cumprod, i, x_0))

    def predict_start_from_noise(self, x_i, i, noise):
        """sde_diffusion.py:220-224."""
        return self._lin(x_i, self._coef(self.sqrt_recip_alphas_cumprod, i, x_i), noise, -self._coef(self.sqrt_recipm1_alphas_cumprod, i, x_i))

    def q_posterior(self, x0, x_i, i):
        """sde_diffusion.py:226-233."""
        mean = self._lin(x0, self._coef(self.posterior_mean_coef1, i, x_i), x_i, self._coef(self.posterior_mean_coef2, i, x_i))
        return (mean, extract(self.posterior_variance, i.to(self.posterior_variance.device), x_i.shape).to(x_i.device),
                extract(self.posterior_log_variance_clipped, i.to(self.posterior_variance.device), x_i.shape).to(x_i.device))

    def p_mean_variance(self, x_start, x, i):
        """sde_diffusion.py:235-237."""
        model_mean, posterior_variance, posterior_log_variance = self.q_posterior(x0=x_start, x_i=x, i=i)
        return model_mean, posterior_variance, posterior_log_variance, x_start

    def q_sample(self, x_start, i):
        """sde_diffusion.py:239-244: returns (x_i, noise); the draw is torch.randn_like, as in the reference."""
        noise = torch.randn_like(x_start)
        return self._lin(x_start, self._coef(self.sqrt_alphas_cumprod, i, x_start), noise,
                         self._coef(self.sqrt_one_minus_alphas_cumprod, i, x_start)), noise

    def host_tables(self):
        """CPU fp32 copies of the buffers (per-step scalars are passed to the kernels by value)."""
        if self._host is None:
            self._host = {n: getattr(self, n).detach().to("cpu", torch.float32).contiguous() for n in TABLE_NAMES}
        return self._host

    def _apply(self, fn, *a, **k):
        self._host = None
        return super()._apply(fn, *a, **k)
