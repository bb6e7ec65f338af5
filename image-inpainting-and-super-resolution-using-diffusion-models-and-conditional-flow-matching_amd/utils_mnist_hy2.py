"""Drop-in for the reference's `mnist/utils_mnist_hy2.py` (imported by mnist/train_mnist_hy2.py:17), MI355X backend.

The ACTIVE generate_samples_eval there (utils_mnist_hy2.py:148-169) is the MNIST 7x7 -> 28x28 super-resolution sampler:
    generate_samples_eval(model, test_images, batch_size=8, step=0, net_="normal") -> (traj, low_res, nfe)
= torchdiffeq dopri5 (atol = rtol = 1e-4) over the tuple state (x, low_res), low_res = bilinear downsample to 7x7,
x_0 ~ N(0, 1) of shape [batch_size, 1, 28, 28]; the result is channel 0 viewed as [-1, 1, 28, 28] and clipped.
"""
from utils_mnist import (_eval_common, device, downsample_images, ema, generate_samples, infiniteloop, use_cuda)  # noqa: F401


def generate_samples_eval(model, test_images, batch_size=8, step=0, net_="normal", *, solver="dopri5", steps=999,
                          image_shape=(1, 28, 28), low_res_size=(7, 7)):
    low_res = downsample_images(test_images, low_res_size).to(device)
    traj, nfe = _eval_common(model, (batch_size, *image_shape), low_res, "low_res", solver, steps)
    return traj, low_res, nfe
