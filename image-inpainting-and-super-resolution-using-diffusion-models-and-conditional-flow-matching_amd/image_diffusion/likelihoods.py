"""Condition builders (the role of `AD/image_diffusion/likelihoods.py:12-158`): in-painting / out-painting masks with the -2
sentinel and the bilinear down-then-up "hyper-resolution" degradation.

They run once per batch, outside the per-step path.  The reference draws one patch position per image in a Python loop
(`Likelihood.sample`, :22-27): the DRAWS are reproduced in that order (they fix the RNG stream); for device tensors the tensor work
of the loop is one HIP launch for the whole batch (`mi355_paint_patch`), and the bilinear resizes are `mi355_resize_bilinear`
(no ATen kernel on device data).  CPU tensors keep the reference's eager expressions (host-side logic, CPU tests).
"""
from typing import Callable, Dict, Type

import torch
import torch.nn.functional as F


def _bilinear(x, size):
    if x.is_cuda:
        from mi355.ops import default_ops

        return default_ops.resize_bilinear(x.float().contiguous(), tuple(size))
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


class Likelihood:
    """`sample(batch)` -> condition tensor; `none_like(x)` -> the "no condition" tensor; `loss(x, condition)` -> data term."""

    def _sample(self, one_image):
        raise NotImplementedError

    def none_like(self, x):
        raise NotImplementedError

    def loss(self, x, condition):
        raise NotImplementedError

    def sample(self, x: torch.Tensor) -> torch.Tensor:
        rows = []
        for k in range(x.shape[0]):          # one independent draw per image, in batch order
            rows.append(self._sample(x[k:k + 1]))
        return torch.cat(rows, dim=0)


class Painting(Likelihood):
    """Square patch of side `patch_size`; `pad_value` marks unknown pixels."""

    def __init__(self, patch_size: int, pad_value: float):
        self.patch_size = patch_size
        self.pad_value = pad_value

    @classmethod
    def from_configdict(cls, config):
        return cls(patch_size=config["patch_size"], pad_value=config["pad_value"])

    def get_random_patch(self, image_size):
        """Top-left corner, at least 5 px from every border (:49-53); two scalar draws, row first."""
        hi = image_size - self.patch_size - 5
        return torch.randint(5, hi, size=()), torch.randint(5, hi, size=())

    def _window(self, images):
        top, left = self.get_random_patch(images.shape[-1])
        return slice(int(top), int(top) + self.patch_size), slice(int(left), int(left) + self.patch_size)

    _outpaint = False

    def sample(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            return super().sample(x)
        # device batch: the same two scalar draws per image, in batch order, then one launch for all images
        from mi355.ops import default_ops

        tops, lefts = [], []
        for _ in range(x.shape[0]):
            top, left = self.get_random_patch(x.shape[-1])
            tops.append(int(top)); lefts.append(int(left))
        t = torch.tensor(tops, dtype=torch.int32).to(x.device)
        l = torch.tensor(lefts, dtype=torch.int32).to(x.device)
        return default_ops.paint_patch(x.detach().float().contiguous(), t, l, self.patch_size, self.pad_value, outpaint=self._outpaint)

    def none_like(self, x):
        return torch.full_like(x, self.pad_value)

    def _known(self, condition):
        return condition != self.pad_value

    def loss(self, x, condition):
        keep = self._known(condition)
        zero = torch.zeros((), dtype=x.dtype, device=x.device)
        diff = torch.where(keep, x, zero) - torch.where(keep, condition, zero)
        return (diff * diff).sum(dim=(1, 2, 3))


class InPainting(Painting):
    """The image with the patch blanked out (:75-87)."""

    def _sample(self, images):
        ys, xs = self._window(images)
        out = images.detach().clone()
        out[:, :, ys, xs] = self.pad_value
        return out


class OutPainting(Painting):
    """Only the patch of the image is known (:90-104)."""

    _outpaint = True

    def _sample(self, images):
        ys, xs = self._window(images)
        out = self.none_like(images)
        out[:, :, ys, xs] = images[:, :, ys, xs].detach()
        return out


class HyperResolution(Likelihood):
    """Bilinear reduction to (target_height, target_width), then bilinear enlargement back (:107-146)."""

    def __init__(self, target_height: int, target_width: int):
        self.target_height = target_height
        self.target_width = target_width

    @classmethod
    def from_configdict(cls, config):
        return cls(config["target_height"], config["target_width"])

    def _sample(self, images):
        small = _bilinear(images, (self.target_height, self.target_width))
        return _bilinear(small, tuple(images.shape[2:4]))

    def sample(self, x):
        if x.is_cuda:            # per-image resizes are independent: the whole batch in two launches
            return self._sample(x)
        return super().sample(x)

    def none_like(self, x):
        return torch.zeros_like(x)

    def loss(self, x, condition):
        return F.mse_loss(_bilinear(condition, x.shape[-2:]), x)


_BY_NAME: Dict[str, Type[Likelihood]] = {"inpainting": InPainting, "outpainting": OutPainting, "hyperresolution": HyperResolution}


def get_likelihood(type_: str) -> Type[Likelihood]:
    cls = _BY_NAME.get(type_.lower())
    if cls is None:
        raise NotImplementedError(f"Unknown conditioning {type_}")
    return cls
