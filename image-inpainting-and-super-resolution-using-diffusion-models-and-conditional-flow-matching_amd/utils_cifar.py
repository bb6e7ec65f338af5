"""Drop-in for the reference's `cifar10/utils_cifar.py` (generate_samples / ema / infiniteloop), MI355X backend.

`generate_samples` keeps the reference signature and file naming (cifar10/utils_cifar.py:13-44): 64 samples,
99 Euler steps over linspace(0, 1, 100), clip(-1,1)/2+0.5, 8x8 PNG grid.  The integration runs inside
libmi355_sampler (one C call), the post-processing is the fused HIP kernel.
"""
import torch

from mi355.imageio import save_image
from mi355.ops import default_ops
from torchcfm_compat import NeuralODE

use_cuda = torch.cuda.is_available()
device = torch.device("cuda" if use_cuda else "cpu")


def _sample_grid(net, n_images=64, n_states=100):
    """64 draws integrated with fixed-step Euler over linspace(0, 1, 100) (99 steps) and mapped to [0, 1]."""
    ode = NeuralODE(net, solver="euler", sensitivity="adjoint")
    start = torch.randn(n_images, 3, 32, 32, device=device)
    times = torch.linspace(0, 1, n_states, device=device)
    with torch.no_grad():
        final = ode.trajectory(start, t_span=times)[-1]
    return default_ops.to_unit_range(final.reshape(-1, 3, 32, 32).contiguous())   # clip(-1, 1) / 2 + 0.5, fused HIP kernel


def generate_samples(model, parallel, savedir, step, net_="normal"):
    """Sanity-check grid written along training (cifar10/utils_cifar.py:13-44): same signature, same file name
    `<savedir><net_>_generated_FM_images_step_<step>.png`, 8 x 8 images; the model is put in eval mode and back in train mode."""
    model.eval()
    # the reference unwraps nn.DataParallel to a single device for torchdyn (utils_cifar.py:30-32)
    net = model.module.to(device) if parallel else model
    grid = _sample_grid(net)
    save_image(grid, f"{savedir}{net_}_generated_FM_images_step_{step}.png", nrow=8)
    model.train()


def _invalidate_engines(target):
    """Drop the packed-weight engine of `target` and of every sub-module that has one: the reference wraps the EMA model in
    nn.DataParallel under --parallel (cifar10/train_cifar10.py:112-113), whose wrapper has no engine of its own - the U-Net sits
    at `.module` - and `generate_samples(ema_model, True, ...)` samples exactly that inner module."""
    seen = set()
    for m in ([target] + list(target.modules() if hasattr(target, "modules") else [])):
        if id(m) not in seen and hasattr(m, "invalidate_engine"):
            seen.add(id(m))
            m.invalidate_engine()


def ema(source, target, decay):
    """cifar10/utils_cifar.py:47-53.  Device-resident fp32 tensors are updated in place by the fused HIP kernel (one launch per
    state-dict entry instead of three eager kernels + a copy); anything else (CPU tensors, integer buffers) keeps the reference's
    eager expression - this is training-side parameter plumbing, not the sampler path."""
    source_dict = source.state_dict()
    target_dict = target.state_dict()
    for key in source_dict.keys():
        t, s = target_dict[key].data, source_dict[key].data
        if t.is_cuda and s.is_cuda and t.dtype == torch.float32 and s.dtype == torch.float32 and t.is_contiguous() and s.is_contiguous():
            default_ops.ema_update_(t, s, decay)
        else:
            t.copy_(t * decay + s * (1 - decay))
    # neither path bumps the parameters' autograd version counters, which the packed-weight cache of UNetModel.engine() keys on:
    # without this, `ema(net, ema_model, d); generate_samples(ema_model, ...)` (cifar10/train_cifar10.py:154-159) would sample
    # from the weights packed at the first engine build
    _invalidate_engines(target)


def infiniteloop(dataloader):
    """Endless stream of the images (labels dropped) of `dataloader` (cifar10/utils_cifar.py:56-59)."""
    import itertools

    for _epoch in itertools.count():
        for batch in dataloader:
            yield batch[0]
