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
