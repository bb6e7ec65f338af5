"""Counterpart of `AD/image_diffusion/sampling.py`: the reverse-denoising samplers.

Same factories and call contracts as the reference:
    get_prior_sample_fn(eps_model, ddpm, conditioning, likelihood)        -> sample(xT)
    get_conditional_sample_fn(eps_model, ddpm, conditioning, likelihood)  -> sample(xT, condition)
`eps_model(xi, i)` takes a long [B] step-index tensor (experiments/main.py:140).  Dispatch is on
`type(conditioning)` (the reference uses the un-vendored `plum` package for that).

Two execution paths, both entirely on the HIP backend:
  * fast path - `eps_model` was made by `make_eps_model(network, ddpm)` with a `UNetModel`: the whole
    loop runs inside libmi355_sampler (`mi355_ddpm_sample`), one C call per sample() call;
  * generic path - any callable `eps_model` (e.g. the reference's own lambda around a UNetModel): a
    host loop that calls it once per step and applies the fused HIP step kernels (csrc/steps.hip).
Noise: by default drawn in-kernel (Philox4x32-10, seeded from torch.initial_seed()); parity tests
inject the draws of the reference's `torch.randn_like` call sequence with `injected_noise([...])`.
ReconstructionGuidance (sampling.py:136-206) differentiates the x0 model through the U-Net: the gradient is the HIP engine's
vector-Jacobian product (`UNetEngine.vjp`, csrc/unet_backward.hip) of a differentiable plan of the same network, so it needs an
`eps_model` made by `make_eps_model(network, ddpm)` around a `UNetModel`.
"""
from __future__ import annotations

import contextlib
import math
from typing import Callable, List, Optional

import torch

from mi355 import _lib
from mi355.ops import default_ops

from .conditioning import Amortized, Conditioning, ReconstructionGuidance, Replacement
from .likelihoods import Likelihood
from .sde_diffusion import DDPM

_ops = default_ops
_injected: Optional[List[torch.Tensor]] = None
_draw_counter = 0


@contextlib.contextmanager
def injected_noise(draws):
    """Use `draws` (list / stacked tensor of [B,C,H,W]) instead of device Philox, in reference call order."""
    global _injected
    prev = _injected
    _injected = list(draws)
    try:
        yield
    finally:
        _injected = prev


@contextlib.contextmanager
def use_ops(ops):
    """Swap the op table (tests pass a recording double to check the host logic without a GPU)."""
    global _ops
    prev = _ops
    _ops = ops
    try:
        yield
    finally:
        _ops = prev


def make_eps_model(network, ddpm: DDPM):
    """eps_model(xi, i) = network(xi, 1.0*i/Ns) (loss_functions.py:18-19, experiments/main.py:140),
    tagged so the samplers can run the whole loop inside the C library."""

    def eps_model(xi, i):
        return network(xi, 1.0 * i / ddpm.Ns)

    eps_model._mi355_network = network
    eps_model._mi355_Ns = ddpm.Ns
    return eps_model


class _Noise:
    """Hands out one noise draw per reference `torch.randn_like` call."""

    def __init__(self, like: torch.Tensor):
        global _draw_counter
        self.like = like
        self.injected = list(_injected) if _injected is not None else None
        self.seed = int(torch.initial_seed()) & ((1 << 63) - 1)
        self.base = _draw_counter
        self.k = 0
        n = like.numel()
        self.stride = (n + 3) // 4 * 4

    def next(self):
        """-> (z tensor or None, philox (seed, offset) or None)"""
        global _draw_counter
        if self.injected is not None:
            if self.k >= len(self.injected):
                raise RuntimeError("injected noise exhausted")
            z = self.injected[self.k].to(self.like.device, torch.float32).contiguous()
            self.k += 1
            return z, None
        off = (self.base + self.k) * self.stride
        self.k += 1
        _draw_counter += 1
        return None, (self.seed, off)


def process_x0(img):
    return _ops.clip_(img, -1.0, 1.0)


def _fast_engine(eps_model, ddpm: DDPM, xT: torch.Tensor):
    net = getattr(eps_model, "_mi355_network", None)
    if net is None or getattr(eps_model, "_mi355_Ns", None) != ddpm.Ns or not hasattr(net, "engine") or not xT.is_cuda:
        return None
    return net.engine(xT.device)


def _none_value(likelihood: Likelihood, like: torch.Tensor) -> float:
    return float(likelihood.pad_value) if hasattr(likelihood, "pad_value") else 0.0


def _tables(ddpm: DDPM):
    return ddpm.host_tables()


def _predictor(eps, xi, i, T, noise: _Noise):
    """step() (sampling.py:59-67): x0_hat -> clip -> posterior mean -> + exp(0.5 logvar) z (z iff i > 0)."""
    z, ph = noise.next() if i > 0 else (None, None)
    sigma = float((0.5 * T["posterior_log_variance_clipped"][i]).exp())  # fp32, like (0.5*logvar).exp()
    _ops.ddpm_step_(xi, eps, z, float(T["sqrt_recip_alphas_cumprod"][i]), float(T["sqrt_recipm1_alphas_cumprod"][i]),
                    float(T["posterior_mean_coef1"][i]), float(T["posterior_mean_coef2"][i]), sigma, ph)
    return xi


def _corrector(eps, xi, i, T, ddpm, delta, noise: _Noise):
    """corrector_step (sampling.py:113-121)."""
    z, ph = noise.next()
    dt = (ddpm.tmax - ddpm.tmin) / ddpm.Ns
    _ops.corrector_step_(xi, eps, z, float(T["sqrt_recip_alphas_cumprod"][i]), float(T["sqrt_recipm1_alphas_cumprod"][i]),
                         float(T["recip_sqrt_m1_alphas_cumprod"][i]), dt, float(delta), ph)
    return xi


def _times(xi, i):
    return torch.full((xi.shape[0],), i, device=xi.device, dtype=torch.long)


def _net_in(xi, cond, amortized):
    return torch.concat((xi, cond), dim=-3) if amortized else xi  # sampling.py:39 (plumbing: the net's own input)


def _run_generic(eps_model, ddpm, xT, *, amortized, cond_pred, cond_corr, replacement=None, n_corrector=0, delta=0.1):
    T = _tables(ddpm)
    xi = xT.detach().clone().float().contiguous()
    noise = _Noise(xi)
    for i in reversed(range(ddpm.Ns)):
        if replacement is not None and i < int(ddpm.Ns * replacement["start_fraction"]):
            z, ph = noise.next() if replacement["noise"] else (None, None)
            _ops.replace_mask_(xi, replacement["condition"], z, replacement["pad_value"], replacement["noise"],
                               float(T["sqrt_alphas_cumprod"][i]), float(T["sqrt_one_minus_alphas_cumprod"][i]), ph)
        eps = eps_model(_net_in(xi, cond_pred, amortized), _times(xi, i))
        _predictor(eps.float().contiguous(), xi, i, T, noise)
        for _ in range(n_corrector):
            eps = eps_model(_net_in(xi, cond_corr, amortized), _times(xi, i))
            _corrector(eps.float().contiguous(), xi, i, T, ddpm, delta, noise)
    return process_x0(xi)


def _run_fast(engine, ddpm, xT, mode, cond, *, n_corrector=0, delta=0.1, start_fraction=1.0, noise_condition=True,
              pad_value=-2.0, none_value=-2.0):
    global _draw_counter
    xi = xT.detach().clone().float().contiguous()
    noise = None
    seed = int(torch.initial_seed()) & ((1 << 63) - 1)
    if _injected is not None:
        noise = torch.stack([z.to(xi.device, torch.float32) for z in _injected]).contiguous()
    else:
        seed = (seed + 0x9E3779B97F4A7C15 * (_draw_counter + 1)) & ((1 << 63) - 1)
        _draw_counter += 1
    engine.ddpm_sample(xi, _tables(ddpm), mode=mode, cond=cond, noise=noise, n_corrector=n_corrector, delta=float(delta),
                       tmin=ddpm.tmin, tmax=ddpm.tmax, start_fraction=float(start_fraction), noise_condition=bool(noise_condition),
                       pad_value=float(pad_value), none_value=float(none_value), seed=seed)
    return xi


# Prior sampling ------------------------------------------------------------------------------------

def get_prior_sample_fn(eps_model: Callable, ddpm: DDPM, conditioning: Conditioning, likelihood: Likelihood):
    """sampling.py:50-75.  Under Amortized conditioning the net is fed `likelihood.none_like` (sampling.py:36-37)."""
    amortized = isinstance(conditioning, Amortized)

    @torch.no_grad()
    def sample(xT):
        engine = _fast_engine(eps_model, ddpm, xT)
        if engine is not None:
            return _run_fast(engine, ddpm, xT, _lib.DDPM_PRIOR, None, none_value=_none_value(likelihood, xT))
        none = likelihood.none_like(xT) if amortized else None
        return _run_generic(eps_model, ddpm, xT, amortized=amortized, cond_pred=none, cond_corr=none)

    return sample


# Conditional sampling ---------------------------------------------------------------------------------

def get_conditional_sample_fn(eps_model: Callable, ddpm: DDPM, conditioning: Conditioning, likelihood: Likelihood):
    if isinstance(conditioning, Amortized):
        return _amortized_sample_fn(eps_model, ddpm, conditioning, likelihood)
    if isinstance(conditioning, Replacement):
        return _replacement_sample_fn(eps_model, ddpm, conditioning, likelihood)
    if isinstance(conditioning, ReconstructionGuidance):
        return _reconstruction_guidance_sample_fn(eps_model, ddpm, conditioning, likelihood)
    raise NotImplementedError(f"no sampler for conditioning type {type(conditioning).__name__}")


def _amortized_sample_fn(eps_model, ddpm, conditioning: Amortized, likelihood):
    """sampling.py:80-133.  The corrector calls the x0 model WITHOUT the condition (sampling.py:116), so the
    network sees none_like there - reproduced."""

    @torch.no_grad()
    def sample(xT, condition):
        condition = condition.to(xT.device).float().contiguous()
        engine = _fast_engine(eps_model, ddpm, xT)
        if engine is not None:
            return _run_fast(engine, ddpm, xT, _lib.DDPM_AMORTIZED, condition, n_corrector=conditioning.n_corrector,
                             delta=conditioning.delta, none_value=_none_value(likelihood, xT))
        return _run_generic(eps_model, ddpm, xT, amortized=True, cond_pred=condition, cond_corr=likelihood.none_like(xT),
                            n_corrector=conditioning.n_corrector, delta=conditioning.delta)

    return sample


def _reconstruction_guidance_sample_fn(eps_model, ddpm, conditioning: ReconstructionGuidance, likelihood):
    """sampling.py:136-206.  Per step i < int(Ns * start_fraction):
        x_grad = vmap(grad(lambda xi: likelihood.loss(x0_model(xi, i), y)))(xi)      (x0_model = clip(predict_start_from_noise(eps_model)))
        x_update = -gamma * alpha_i * (1 - alpha_i) * x_grad;  "before": xi += x_update (the predictor then runs on the moved xi),
        "after": the predictor runs on xi and x_update is added to its result.
    The losses are per sample and the network has no cross-sample coupling, so vmap(grad) is ONE batched backward pass:
    seed kernel (clip / loss / predict_start chain rule) -> UNetEngine.vjp -> update kernel."""
    net = getattr(eps_model, "_mi355_network", None)
    if net is None or getattr(eps_model, "_mi355_Ns", None) != ddpm.Ns or not hasattr(net, "engine"):
        raise NotImplementedError(
            "ReconstructionGuidance differentiates through the network: pass an eps_model built by make_eps_model(network, ddpm) around "
            "a UNetModel (an arbitrary callable has no backward pass on the HIP backend)")
    if conditioning.update_rule not in ("before", "after"):
        raise ValueError(f"unknown update_rule {conditioning.update_rule!r}")
    if hasattr(likelihood, "pad_value"):
        mode, pad = 0, float(likelihood.pad_value)      # Painting.loss (likelihoods.py:58-66)
    elif type(likelihood).__name__ == "HyperResolution":
        mode, pad = 1, 0.0                              # HyperResolution.loss (likelihoods.py:138-143)
    else:
        raise NotImplementedError(f"no constraint gradient for likelihood {type(likelihood).__name__}")

    @torch.no_grad()
    def sample(xT, condition):
        if not xT.is_cuda:
            from mi355._lib import MI355BackendError
            raise MI355BackendError("ReconstructionGuidance sampling needs device tensors (no CPU fallback)")
        T = _tables(ddpm)
        alphas = ddpm.alphas.detach().to("cpu", torch.float32)
        condition = condition.to(xT.device).float().contiguous()
        xi = xT.detach().clone().float().contiguous()
        B = xi.shape[0]
        deng, eng = net.engine(xi.device, differentiable=True), net.engine(xi.device)
        noise = _Noise(xi)
        n_guided = int(ddpm.Ns * conditioning.start_fraction)
        for i in reversed(range(ddpm.Ns)):
            t = torch.full((B,), 1.0 * i / ddpm.Ns, device=xi.device, dtype=torch.float32)
            update, eps = None, None
            if i < n_guided:
                eps = deng.forward(xi, t)
                g_eps, g_x = _ops.guidance_seed(xi, eps, condition, float(T["sqrt_recip_alphas_cumprod"][i]),
                                                float(T["sqrt_recipm1_alphas_cumprod"][i]), mode, pad)
                vjp = deng.vjp(g_eps)
                a_i = float(alphas[i])
                scale = float(conditioning.gamma) * a_i * (1.0 - a_i)
                before = conditioning.update_rule == "before"
                update = _ops.guidance_update_(xi, g_x, vjp, scale, before)
                if before:
                    eps = None                       # xi moved: the predictor needs the network at the new point
            if eps is None:
                eps = eng.forward(xi, t)
            _predictor(eps, xi, i, T, noise)
            if update is not None and conditioning.update_rule == "after":
                _ops.euler_step_(xi, update, 1.0)    # pred_img += x_update
            for _ in range(conditioning.n_corrector):
                epc = eng.forward(xi, t)
                _corrector(epc, xi, i, T, ddpm, conditioning.delta, noise)
        return process_x0(xi)

    return sample


def _replacement_sample_fn(eps_model, ddpm, conditioning: Replacement, likelihood):
    """sampling.py:209-260: overwrite the known pixels with the (noised) condition before each predictor step."""

    @torch.no_grad()
    def sample(xT, condition):
        condition = condition.to(xT.device).float().contiguous()
        pad = float(likelihood.pad_value)
        engine = _fast_engine(eps_model, ddpm, xT)
        if engine is not None:
            return _run_fast(engine, ddpm, xT, _lib.DDPM_REPLACEMENT, condition, n_corrector=conditioning.n_corrector,
                             delta=conditioning.delta, start_fraction=conditioning.start_fraction,
                             noise_condition=conditioning.noise, pad_value=pad)
        rep = dict(condition=condition, start_fraction=conditioning.start_fraction, noise=bool(conditioning.noise), pad_value=pad)
        return _run_generic(eps_model, ddpm, xT, amortized=False, cond_pred=None, cond_corr=None, replacement=rep,
                            n_corrector=conditioning.n_corrector, delta=conditioning.delta)

    return sample


def get_ddim_sample_fn(eps_model: Callable, ddpm: DDPM, likelihood: Optional[Likelihood] = None):
    """BUILD-DEFINED EXTENSION (no reference counterpart; BASELINE.json configs 3/5 name DDIM): deterministic
    DDIM(eta=0) on the same tables with the reference's clipped x0_hat.  sample(xT, condition=None)."""

    @torch.no_grad()
    def sample(xT, condition=None):
        T = _tables(ddpm)
        engine = _fast_engine(eps_model, ddpm, xT)
        if condition is not None:
            condition = condition.to(xT.device).float().contiguous()
        if engine is not None:
            nv = _none_value(likelihood, xT) if likelihood is not None else 0.0
            return _run_fast(engine, ddpm, xT, _lib.DDIM, condition, none_value=nv)
        xi = xT.detach().clone().float().contiguous()
        for i in reversed(range(ddpm.Ns)):
            eps = eps_model(_net_in(xi, condition, condition is not None), _times(xi, i))
            _ops.ddim_step_(xi, eps.float().contiguous(), float(T["sqrt_recip_alphas_cumprod"][i]),
                            float(T["sqrt_recipm1_alphas_cumprod"][i]), float(T["alphas_cumprod_prev"][i]))
        return process_x0(xi)

    return sample
