"""fp32 restatement of the CFM ODE-integration sampler and its post-processing (TEST ORACLE).

The fixed-step Euler solver lives in the un-vendored `torchdyn` package (version unpinned by the
reference; weight URLs in cifar10/README.md:45-49 point at torchcfm release 1.0.4).  Its published
algorithm for `NeuralODE(model, solver="euler").trajectory(x, t_span)` is restated here and anchored
on the reference's call sites: cifar10/compute_fid.py:69-79,86-87, cifar10/utils_cifar.py:34-41,
mnist/utils_mnist2.py:118-138.  No reference test or golden vector pins it => "parity unpinned"
for the loop; the U-Net inside it is pinned (tests/golden/unet_*.npz).
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn.functional as F


@torch.no_grad()
def euler_trajectory(f: Callable, x: torch.Tensor, t_span: torch.Tensor, keep_all: bool = True) -> torch.Tensor:
    """x_{k+1} = x_k + (t_{k+1} - t_k) f(t_k, x_k); returns the stack of all len(t_span) states
    (torchdyn fixed-step odeint semantics) or only the last one."""
    xs = [x]
    for k in range(len(t_span) - 1):
        t, dt = t_span[k], t_span[k + 1] - t_span[k]
        x = x + dt * f(t, x)
        if keep_all:
            xs.append(x)
    return torch.stack(xs) if keep_all else x


@torch.no_grad()
def euler_concat_state(model: Callable, x0: torch.Tensor, con: torch.Tensor, t_span: torch.Tensor):
    """mnist/utils_mnist2.py:118-138, restated: the ODE state is torch.cat((x, con), dim=1) and
    ode_func(t, s) = cat(model(s_x, t, con=s_con), s_con) - the condition half's derivative is the condition itself, so under
    fixed-step Euler the model is fed con_k = con * prod(1 + dt_j).  model(x, t, con) -> dx/dt.  Returns (x_final, con_final)."""
    s = torch.cat((x0, con), dim=1)
    C = x0.shape[1]
    for k in range(len(t_span) - 1):
        t, dt = t_span[k], t_span[k + 1] - t_span[k]
        ds = torch.cat((model(s[:, :C], t, s[:, C:]), s[:, C:]), dim=1)
        s = s + dt * ds
    return s[:, :C], s[:, C:]


def to_uint8(x: torch.Tensor) -> torch.Tensor:
    """cifar10/compute_fid.py:87: (x*127.5 + 128).clip(0, 255).to(uint8)  (truncation toward zero)."""
    return (x * 127.5 + 128).clip(0, 255).to(torch.uint8)


def to_unit_range(x: torch.Tensor) -> torch.Tensor:
    """cifar10/utils_cifar.py:40-41: clip(-1, 1) / 2 + 0.5."""
    return x.clip(-1, 1) / 2 + 0.5


@torch.no_grad()
def gen_images_u8(f: Callable, x0: torch.Tensor, steps: int) -> torch.Tensor:
    """gen_1_img, Euler branch (cifar10/compute_fid.py:73-88) with the initial draw x0 injected."""
    t_span = torch.linspace(0, 1, steps + 1)
    return to_uint8(euler_trajectory(f, x0, t_span, keep_all=False))


# --- condition builders -----------------------------------------------------------------------

def inpainting_condition(images: torch.Tensor, h: int, w: int, patch: int, pad_value: float = -2.0) -> torch.Tensor:
    """InPainting._sample (likelihoods.py:78-87) with the patch corner (h, w) injected; applies the
    same corner to every image of the batch slice it is given (the reference calls it per image)."""
    cond = images.detach().clone()
    cond[:, :, h:h + patch, w:w + patch] = pad_value
    return cond


def outpainting_condition(images: torch.Tensor, h: int, w: int, patch: int, pad_value: float = -2.0) -> torch.Tensor:
    """OutPainting._sample (likelihoods.py:95-104)."""
    cond = torch.ones_like(images) * pad_value
    cond[:, :, h:h + patch, w:w + patch] = images[:, :, h:h + patch, w:w + patch]
    return cond


def hyperresolution_condition(images: torch.Tensor, th: int, tw: int) -> torch.Tensor:
    """HyperResolution._sample (likelihoods.py:119-126): bilinear down (align_corners=False) then up."""
    low = F.interpolate(images, size=(th, tw), mode="bilinear", align_corners=False)
    return F.interpolate(low, (images.shape[2], images.shape[3]), mode="bilinear")


def downsample_images(images: torch.Tensor, target_size) -> torch.Tensor:
    """mnist/utils_mnist_hy.py:18-28."""
    return F.interpolate(images, size=target_size, mode="bilinear", align_corners=False)


def painting_loss(x: torch.Tensor, condition: torch.Tensor, pad_value: float = -2.0) -> torch.Tensor:
    """Painting.loss (likelihoods.py:58-66)."""
    x = torch.where(condition == pad_value, 0.0, x)
    c = torch.where(condition == pad_value, 0.0, condition)
    return torch.sum((x - c) ** 2, dim=(1, 2, 3))


# --- adaptive Dormand-Prince 5(4): torchdiffeq.odeint(method="dopri5") restated (un-vendored, version unpinned) ------------
_DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_BETA = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
            [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656], [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
_DP_CERR = [35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 - -12231 / 42400,
            11 / 84 - 649 / 6300, -1.0 / 60.0]
_DP_CMID = [6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
            187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


@torch.no_grad()
def dopri5(func, y0, t0: float, t_end: float, rtol: float, atol: float):
    """odeint(func, y0, [t0, t_end], rtol, atol, method="dopri5")[-1] as called at cifar10/compute_fid.py:80-85 and
    mnist/utils_mnist.py:101-108.  `y0` is a tensor or a tuple of tensors (tuple states use the max of the per-component
    RMS norms).  Returns (y(t_end), nfe).  Published torchdiffeq 0.2.x algorithm; no reference test pins it."""
    tup = isinstance(y0, (tuple, list))
    ys = [v.float() for v in (y0 if tup else [y0])]
    nfe = [0]

    def f(t, y):
        nfe[0] += 1
        out = func(torch.tensor(t), tuple(y)) if tup else [func(torch.tensor(t), y[0])]
        return [o.float() for o in out]

    def norm(vs):
        return max(float(v.double().pow(2).mean().sqrt()) for v in vs)

    def comb(base, ks, cs):
        out = []
        for i in range(len(ys)):
            acc = torch.zeros_like(ks[0][i])
            for k, c in zip(ks, cs):
                acc = acc + k[i] * c
            out.append(acc if base is None else base[i] + acc)
        return out

    f0 = f(t0, ys)
    scale = [atol + v.abs() * rtol for v in ys]
    d0 = norm([v / s for v, s in zip(ys, scale)])
    d1 = norm([v / s for v, s in zip(f0, scale)])
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = f(t0 + h0, comb(ys, [f0], [h0]))
    d2 = norm([(a - b) / s for a, b, s in zip(f1, f0, scale)]) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** 0.2
    dt = min(100 * h0, h1)
    t, interp = t0, None
    while t_end > t:
        t1 = t + dt
        ks = [f0]
        yi = None
        for a, beta in zip(_DP_ALPHA, _DP_BETA):
            yi = comb(ys, ks, [b * dt for b in beta])
            ks.append(f(t1 if a == 1.0 else t + a * dt, yi))
        y1, f1 = yi, ks[-1]
        err = comb(None, ks, [c * dt for c in _DP_CERR])
        ratio = norm([e / (atol + rtol * torch.max(a.abs(), b.abs())) for e, a, b in zip(err, ys, y1)])
        if ratio <= 1.0:
            interp = (ys, y1, comb(ys, ks, [c * dt for c in _DP_CMID]), f0, f1, t, dt)
            t, ys, f0 = t1, y1, f1
        factor = 10.0 if ratio == 0.0 else min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1.0 else 0.2))
        dt = dt * factor
    ya, yb, ym, fa, fb, ta, dta = interp
    x = (t_end - ta) / dta
    out = []
    for i in range(len(ya)):
        a = 2 * dta * (fb[i] - fa[i]) - 8 * (yb[i] + ya[i]) + 16 * ym[i]
        b = dta * (5 * fa[i] - 3 * fb[i]) + 18 * ya[i] + 14 * yb[i] - 32 * ym[i]
        c = dta * (fb[i] - 4 * fa[i]) - 11 * ya[i] - 5 * yb[i] + 16 * ym[i]
        out.append((((a * x + b) * x + c) * x + dta * fa[i]) * x + ya[i])
    return (tuple(out) if tup else out[0]), nfe[0]
