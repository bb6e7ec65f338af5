"""Drop-in for the reference's `mnist/utils_mnist*.py` sampler helpers, MI355X backend (fixed-step Euler).

Kept: get_random_patch / _sample / sample (utils_mnist.py:16-41), downsample_images (utils_mnist_hy.py:18-28),
generate_samples (utils_mnist.py:45-73), generate_samples_eval in its Euler form (utils_mnist2.py:118-138, the
active definition there: 999 Euler steps over linspace(0,1,1000), state = channel-concat [x, con]) for in-painting
(`con=`) and its super-resolution sibling (`low_res=`, utils_mnist_hy.py:76-98), ema, infiniteloop.
The reference's dopri5 variants (utils_mnist.py:90-134, utils_mnist_hy*.py) run on mi355.ode.Dopri5 (the torchdiffeq
algorithm restated, HIP kernels for stage combination / error norm / dense output), including the tuple-state quirk;
`solver="euler", steps=N` selects the fixed-step form.
"""
import torch
import torch.nn.functional as F

from mi355.imageio import save_image
from mi355.ops import default_ops
from torchcfm_compat import InPaintModelWrapper, NeuralODE, SuperResModelWrapper

use_cuda = torch.cuda.is_available()
device = torch.device("cuda" if use_cuda else "cpu")


def get_random_patch(image_size=28, patch_size=14):
    # don't sample too close to the border (utils_mnist.py:16-20)
    h = torch.randint(5, image_size - patch_size - 5, size=())
    w = torch.randint(5, image_size - patch_size - 5, size=())
    return h, w


def _sample(images, pad_value=2, patch_size=14):
    """images: [N, C, H, W] -> copy with a 14x14 patch set to -2 (utils_mnist.py:23-34; the arguments are
    overridden inside the reference function, reproduced)."""
    image_size = images.shape[-1]
    pad_value = -2
    patch_size = 14
    h, w = get_random_patch(image_size, patch_size)
    condition = images.detach().clone()
    condition[:, :, h:h + patch_size, w:w + patch_size] = pad_value
    return condition


def sample(x):
    return torch.cat([_sample(x[[k]]) for k in range(x.shape[0])], dim=0)


def downsample_images(images, target_size):
    """utils_mnist_hy.py:18-28."""
    return F.interpolate(images, size=target_size, mode="bilinear", align_corners=False)


def _check_solver(solver):
    if solver not in ("euler", "dopri5"):
        raise NotImplementedError(f"solver={solver!r}: only 'euler' and 'dopri5' are built")


def generate_samples(model, parallel, savedir, step, net_="normal", solver="dopri5", steps=99, image_shape=(1, 28, 28)):
    """utils_mnist.py:45-73: NeuralODE(model, solver="dopri5", atol=1e-4, rtol=1e-4) over linspace(0, 1, 100), 64 samples."""
    _check_solver(solver)
    model.eval()
    node_ = NeuralODE(model, solver=solver, sensitivity="adjoint", atol=1e-4, rtol=1e-4)
    with torch.no_grad():
        traj = node_.trajectory(torch.randn(64, *image_shape, device=device), t_span=torch.linspace(0, 1, steps + 1, device=device))
        traj = default_ops.to_unit_range(traj[-1, :].view([-1, *image_shape]).contiguous())
    save_image(traj, savedir + f"{net_}_generated_FM_images_step_{step}.png", nrow=8)
    model.train()


def ema(source, target, decay):
    source_dict = source.state_dict()
    target_dict = target.state_dict()
    for key in source_dict.keys():
        target_dict[key].data.copy_(target_dict[key].data * decay + source_dict[key].data * (1 - decay))


def infiniteloop(dataloader):
    while True:
        for x, y in iter(dataloader):
            yield x


def _euler_conditional(model, x_0, cond, steps):
    """x_{k+1} = x_k + dt * model(x_k, t_k, cond); the condition is carried unchanged (utils_mnist2.py:120-124:
    the concatenated state's second half has derivative `con` under torchdyn there, but only x[:,0] is read)."""
    ts = torch.linspace(0, 1, steps + 1).tolist()
    x = x_0.detach().clone().float().contiguous()
    if isinstance(model, (InPaintModelWrapper, SuperResModelWrapper)) and x.is_cuda:
        c = cond if isinstance(model, InPaintModelWrapper) else F.interpolate(cond, (x.shape[2], x.shape[3]), mode="bilinear")
        model.engine(x.device).cfm_euler(x, ts, cond=c.float().contiguous())
        return x, steps
    kw = "con" if not isinstance(model, SuperResModelWrapper) else "low_res"
    for k in range(steps):
        t = torch.tensor(ts[k], device=x.device)
        v = model.forward(x, t, **{kw: cond})
        default_ops.euler_step_(x, v.float().contiguous(), ts[k + 1] - ts[k])
    return x, steps


def _dopri5_conditional(model, x_0, cond, kw):
    """torchdiffeq.odeint over the TUPLE state (x, cond) with ode_func = (model.forward(x, t, <kw>=cond), cond), atol = rtol =
    1e-4 (utils_mnist.py:95-108, utils_mnist_hy.py:79-92).  Reference quirk, reproduced: the second component's derivative is
    the condition itself, so the condition the model sees drifts like e^t along the solve, and it takes part in the error norm."""
    from mi355.ode import odeint_dopri5

    def f(t, st):
        tt = torch.tensor(float(t), device=st[0].device)
        return (model.forward(st[0], tt, **{kw: st[1]}), st[1])

    (x, _), nfe = odeint_dopri5(f, (x_0.float().contiguous(), cond.float().contiguous()), 0.0, 1.0, 1e-4, 1e-4)
    return x, nfe


def generate_samples_eval(model, test_images, savedir=None, batch_size=8, step=0, net_="normal", solver="dopri5", steps=999):
    """In-painting evaluation sampler: (traj, con, nfe).  solver="dopri5" is utils_mnist.py:90-134 (tuple state, adaptive);
    solver="euler" is utils_mnist2.py:118-138 (999 Euler steps)."""
    _check_solver(solver)
    model.eval()
    with torch.no_grad():
        con = sample(test_images).to(device)
        x_0 = torch.randn(batch_size, *test_images.shape[1:], device=device)
        if solver == "dopri5":
            x, nfe = _dopri5_conditional(model, x_0, con, "con")
        else:
            x, nfe = _euler_conditional(model, x_0, con.float().contiguous(), steps)
        traj = default_ops.clip_(x.view([-1, *test_images.shape[1:]]).contiguous(), -1.0, 1.0)
    model.train()
    return traj, con, nfe


def generate_samples_eval_superres(model, test_images, batch_size=8, step=0, net_="normal", low_res_size=(16, 16), solver="dopri5",
                                   steps=100):
    """Super-resolution evaluation sampler (utils_mnist_hy.py:76-98, utils_mnist_hy2.py:148-169): (traj, low_res, nfe)."""
    _check_solver(solver)
    model.eval()
    with torch.no_grad():
        low_res = downsample_images(test_images, low_res_size).to(device)
        x_0 = torch.randn(batch_size, *test_images.shape[1:], device=device)
        if solver == "dopri5":
            x, nfe = _dopri5_conditional(model, x_0, low_res, "low_res")
        else:
            x, nfe = _euler_conditional(model, x_0, low_res.float().contiguous(), steps)
        traj = default_ops.clip_(x.view([-1, *test_images.shape[1:]]).contiguous(), -1.0, 1.0)
    model.train()
    return traj, low_res, nfe
