"""MI355X-native drop-in for the reference package `image_diffusion` (sampling hot path only).

Same module / symbol names as `amortised diffusion/image_diffusion/` for the path SURVEY.md section 8
scopes: nn, unet, sde_diffusion, conditioning, likelihoods, sampling.  Training, logging, datasets
and plotting modules of the reference are out of scope and not provided.
"""
from . import conditioning, likelihoods, nn, sampling, sde_diffusion, unet  # noqa: F401
