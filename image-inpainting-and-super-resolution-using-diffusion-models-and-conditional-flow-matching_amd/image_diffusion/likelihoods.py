"""Counterpart of `AD/image_diffusion/likelihoods.py:12-158`: builders of the condition tensor.

These run once per batch on the host side of the sampler (the reference loops over images in
Python); they only index / fill / resize tensors with PyTorch and never touch the per-step path.
"""
from typing import Type

import numpy as np
import torch
import torch.nn.functional as F


class Likelihood:
    def sample(self, x: torch.Tensor) -> torch.Tensor:
        # likelihoods.py:29-34: one independent draw per image
        return torch.cat([self._sample(x[[k]]) for k in range(x.shape[0])], dim=0)

    def _sample(self, x):
        raise NotImplementedError

    def none_like(self, x):
        raise NotImplementedError

    def loss(self, x, condition):
        raise NotImplementedError


class Painting(Likelihood):
    @classmethod
    def from_configdict(cls, config):
        return cls(patch_size=config["patch_size"], pad_value=config["pad_value"])

    def __init__(self, patch_size: int, pad_value: float):
        self.pad_value, self.patch_size = pad_value, patch_size

    def get_random_patch(self, image_size):
        # likelihoods.py:49-53: keep 5 px from the border
        h = torch.randint(5, image_size - self.patch_size - 5, size=())
        w = torch.randint(5, image_size - self.patch_size - 5, size=())
        return h, w

    def none_like(self, x):
        return torch.ones_like(x) * self.pad_value

    def loss(self, x, condition):
        x = torch.where(condition == self.pad_value, 0.0, x)
        condition = torch.where(condition == self.pad_value, 0.0, condition)
        return torch.sum((x - condition) ** 2, dim=(1, 2, 3))


class InPainting(Painting):
    """Condition = image with a patch set to the sentinel (likelihoods.py:75-87)."""

    def _sample(self, images):
        h, w = self.get_random_patch(images.shape[-1])
        condition = images.detach().clone()
        condition[np.s_[:, :, h:h + self.patch_size, w:w + self.patch_size]] = self.pad_value
        return condition


class OutPainting(Painting):
    """Condition = sentinel everywhere except a patch of the image (likelihoods.py:90-104)."""

    def _sample(self, images):
        h, w = self.get_random_patch(images.shape[-1])
        s = np.s_[:, :, h:h + self.patch_size, w:w + self.patch_size]
        condition = torch.ones_like(images) * self.pad_value
        condition[s] = images[s].detach().clone()
        return condition


class HyperResolution(Likelihood):
    """Bilinear down to (th, tw) then back up (likelihoods.py:107-146)."""

    @classmethod
    def from_configdict(cls, config):
        return cls(config["target_height"], config["target_width"])

    def __init__(self, target_height: int, target_width: int):
        self.target_height, self.target_width = target_height, target_width

    def _sample(self, images):
        low = F.interpolate(images, size=(self.target_height, self.target_width), mode="bilinear", align_corners=False)
        return F.interpolate(low, (images.shape[2], images.shape[3]), mode="bilinear")

    def none_like(self, x):
        return torch.zeros_like(x)

    def loss(self, x, condition):
        up = F.interpolate(condition, size=x.shape[-2:], mode="bilinear", align_corners=False)
        return F.mse_loss(up, x)


def get_likelihood(type_: str) -> Type[Likelihood]:
    t = type_.lower()
    if t == "inpainting":
        return InPainting
    if t == "outpainting":
        return OutPainting
    if t == "hyperresolution":
        return HyperResolution
    raise NotImplementedError(f"Unknown conditioning {type_}")
