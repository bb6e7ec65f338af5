"""Drop-in for the reference's `mnist/utils_mnist_hy.py` (imported by mnist/train_mnist_hy.py:17), MI355X backend.

The ACTIVE generate_samples_eval there (utils_mnist_hy.py:76-98) is the 64x64 super-resolution sampler:
    generate_samples_eval(model, test_images, batch_size=8, step=0, net_="normal") -> (traj, low_res, nfe)
= torchdiffeq dopri5 (atol = rtol = 1e-4) over the tuple state (x, low_res), low_res = bilinear downsample of the test
images to 16x16, x_0 ~ N(0, 1) of shape [batch_size, 3, 64, 64]; the model is called as model.forward(x, t, low_res=...)
(SuperResModelWrapper upsamples and concatenates it).  The tuple-state quirk (low_res drifts like e^t) is reproduced.
"""
from utils_mnist import (_eval_common, device, downsample_images, ema, generate_samples, infiniteloop, use_cuda)  # noqa: F401


def generate_samples_eval(model, test_images, batch_size=8, step=0, net_="normal", *, solver="dopri5", steps=99,
                          image_shape=(3, 64, 64), low_res_size=(16, 16)):
    low_res = downsample_images(test_images, low_res_size).to(device)
    traj, nfe = _eval_common(model, (batch_size, *image_shape), low_res, "low_res", solver, steps)
    return traj, low_res, nfe
