"""Drop-in for the reference's `mnist/utils_mnist2.py` (imported by mnist/train_mnist2.py:17), MI355X backend.

Differences from utils_mnist, as in the reference: the mask patch is 20 pixels (utils_mnist2.py:29) and the ACTIVE
generate_samples_eval (utils_mnist2.py:118-138) is the fixed-step one:
    generate_samples_eval(model, test_images, batch_size=8, step=0, net_="normal") -> (traj, con, nfe)
= torchdyn NeuralODE(solver="euler") over linspace(0, 1, 1000) (999 steps, nfe = 999) on the channel-concatenated state
[x, con]; the condition half has derivative con, so the model is fed a drifting condition (see utils_mnist._euler_conditional).
The whole loop is one C call (mi355_cfm_euler_sample with cond_drift).
"""
import torch

from utils_mnist import (_eval_common, _sample_patch, device, ema, generate_samples, get_random_patch, infiniteloop,  # noqa: F401
                         use_cuda)


def _sample(images, pad_value=2, patch_size=14):
    """utils_mnist2.py:23-34: both arguments are overridden inside the reference function (pad -2, patch 20) - reproduced."""
    return _sample_patch(images, 20)


def sample(x):
    return torch.cat([_sample(x[[k]]) for k in range(x.shape[0])], dim=0)


def generate_samples_eval(model, test_images, batch_size=8, step=0, net_="normal", *, solver="euler", steps=999,
                          image_shape=(1, 28, 28)):
    con = sample(test_images).to(device)
    traj, nfe = _eval_common(model, (batch_size, *image_shape), con, "con", solver, steps)
    return traj, con, nfe
