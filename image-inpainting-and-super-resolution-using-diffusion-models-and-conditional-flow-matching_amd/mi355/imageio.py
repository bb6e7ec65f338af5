"""PNG grid writer with torchvision.utils.save_image / make_grid semantics (torchvision is not installed here).

Restates make_grid(nrow, padding=2, pad_value=0) + save_image's `mul(255).add_(0.5).clamp_(0, 255).to(uint8)`;
used by utils_cifar / utils_mnist `generate_samples` (cifar10/utils_cifar.py:42).  Host-side file I/O only.
"""
from __future__ import annotations

import math

import torch


def make_grid(tensor: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    if tensor.dim() == 3:
        tensor = tensor.unsqueeze(0)
    if tensor.shape[1] == 1:
        tensor = tensor.expand(-1, 3, -1, -1)
    nmaps = tensor.shape[0]
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.shape[2] + padding), int(tensor.shape[3] + padding)
    grid = tensor.new_full((tensor.shape[1], height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid[:, y * height + padding:(y + 1) * height, x * width + padding:(x + 1) * width] = tensor[k]
            k += 1
    return grid


def save_image(tensor: torch.Tensor, fp: str, nrow: int = 8, padding: int = 2) -> None:
    from PIL import Image

    grid = make_grid(tensor.detach().float().cpu(), nrow=nrow, padding=padding)
    ndarr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(ndarr).save(fp)
