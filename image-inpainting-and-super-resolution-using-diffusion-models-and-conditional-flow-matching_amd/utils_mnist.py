"""Drop-in for the reference's `mnist/utils_mnist.py` (imported by mnist/train_mnist.py:17), MI355X backend.

Same names and call contracts: get_random_patch / _sample / sample (utils_mnist.py:16-41, 14-pixel patch), generate_samples
(:45-73, dopri5, 64 samples of 1x28x28, PNG grid), ema / infiniteloop (:76-88) and
    generate_samples_eval(model, test_images, savedir, batch_size=8, step=0, net_="normal") -> (traj, con)     (:90-135)
= torchdiffeq dopri5 (atol = rtol = 1e-4) over the TUPLE state (x, con).  The adaptive solver is mi355.ode (the torchdiffeq
algorithm restated over HIP kernels: stage combination / error norm / dense output), including the tuple-state quirk.
The sibling modules utils_mnist2 / utils_mnist_hy / utils_mnist_hy2 mirror the other three reference files (each has its own
generate_samples_eval signature and return tuple); the shared sampler bodies live here (`_euler_conditional`,
`_dopri5_conditional`).  Keyword-only extensions (`solver=`, `steps=`, `image_shape=`) select the other solver / a different
state shape without changing the positional contract.
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


def _sample_patch(images, patch_size):
    image_size = images.shape[-1]
    h, w = get_random_patch(image_size, patch_size)
    condition = images.detach().clone()
    condition[:, :, h:h + patch_size, w:w + patch_size] = -2
    return condition


def _sample(images, pad_value=2, patch_size=14):
    """images: [N, C, H, W] -> copy with a 14x14 patch set to -2 (utils_mnist.py:23-34; both arguments are
    overridden inside the reference function - reproduced)."""
    return _sample_patch(images, 14)


def sample(x):
    return torch.cat([_sample(x[[k]]) for k in range(x.shape[0])], dim=0)


def downsample_images(images, target_size):
    """utils_mnist_hy.py:18-28: F.interpolate(images, size, mode="bilinear", align_corners=False); device tensors go through the HIP
    resize kernel, CPU tensors (data-loader side) keep the eager call."""
    if images.is_cuda:
        return default_ops.resize_bilinear(images.float().contiguous(), tuple(target_size))
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
    """utils_mnist.py:76-82.  `.data.copy_` does not bump the autograd version counters the packed-weight cache keys on,
    so the target's engine is invalidated explicitly."""
    source_dict = source.state_dict()
    target_dict = target.state_dict()
    for key in source_dict.keys():
        target_dict[key].data.copy_(target_dict[key].data * decay + source_dict[key].data * (1 - decay))
    _invalidate_engines(target)


def infiniteloop(dataloader):
    while True:
        for x, y in iter(dataloader):
            yield x


def _euler_conditional(model, x_0, cond, steps, drift=True):
    """Fixed-step Euler over the reference's CONCATENATED state [x, con] (utils_mnist2.py:118-138): ode_func returns
    cat(model(x, t, con=con), con), so the second half's derivative is con itself and the condition the model is fed drifts,
    con_{k+1} = con_k + dt * con_k (about e^t * con; the -2 sentinel reaches about -5.4).  Reproduced (drift=True); the
    caller's `cond` tensor is not modified.  -> (x_1, nfe)"""
    ts = torch.linspace(0, 1, steps + 1).tolist()
    x = x_0.detach().clone().float().contiguous()
    if isinstance(model, (InPaintModelWrapper, SuperResModelWrapper)) and x.is_cuda:
        # bilinear up-sampling is linear, so the drifting low-res condition (con_{k+1} = (1 + dt) con_k) may be up-sampled ONCE per
        # solve and drifted at full size: up((1 + dt) c) = (1 + dt) up(c)
        c = cond if isinstance(model, InPaintModelWrapper) else model.upsample(cond, (x.shape[2], x.shape[3]))
        model.engine(x.device).cfm_euler(x, ts, cond=c.float().contiguous(), cond_drift=drift)
        return x, steps
    kw = "con" if not isinstance(model, SuperResModelWrapper) else "low_res"
    c = cond.detach().clone().float().contiguous()
    for k in range(steps):
        t = ts[k] if isinstance(model, (InPaintModelWrapper, SuperResModelWrapper)) else torch.tensor(ts[k], device=x.device)
        v = model.forward(x, t, **{kw: c})
        default_ops.euler_step_(x, v.float().contiguous(), ts[k + 1] - ts[k])
        if drift:
            default_ops.euler_step_(c, c, ts[k + 1] - ts[k])
    return x, steps


def _dopri5_conditional(model, x_0, cond, kw):
    """torchdiffeq.odeint over the TUPLE state (x, cond) with ode_func = (model.forward(x, t, <kw>=cond), cond), atol = rtol =
    1e-4 (utils_mnist.py:95-108, utils_mnist_hy.py:79-92).  Reference quirk, reproduced: the second component's derivative is
    the condition itself, so the condition the model sees drifts like e^t along the solve, and it takes part in the error norm."""
    from mi355.ode import odeint_dopri5

    host_t = isinstance(model, (InPaintModelWrapper, SuperResModelWrapper))   # the wrappers take the solver's host scalar as it is

    def f(t, st):
        tt = float(t) if host_t else torch.tensor(float(t), device=st[0].device)
        return (model.forward(st[0], tt, **{kw: st[1]}), st[1])

    (x, _), nfe = odeint_dopri5(f, (x_0.float().contiguous(), cond.float().contiguous()), 0.0, 1.0, 1e-4, 1e-4)
    return x, nfe


def _eval_common(model, x_shape, cond, kw, solver, steps):
    """Shared body of the four generate_samples_eval variants: x_0 ~ N(0, 1), integrate, clip(-1, 1).  -> (traj, nfe)"""
    _check_solver(solver)
    model.eval()
    with torch.no_grad():
        x_0 = torch.randn(*x_shape, device=device)
        if solver == "dopri5":
            x, nfe = _dopri5_conditional(model, x_0, cond, kw)
        else:
            x, nfe = _euler_conditional(model, x_0, cond.float().contiguous(), steps)
        traj = default_ops.clip_(x.view([-1, *x_shape[1:]]).contiguous(), -1.0, 1.0)
    model.train()
    return traj, nfe


def generate_samples_eval(model, test_images, savedir, batch_size=8, step=0, net_="normal", *, solver="dopri5", steps=999,
                          image_shape=(1, 28, 28)):
    """utils_mnist.py:90-135: in-painting evaluation sampler, dopri5 over the tuple state (x, con); returns (traj, con) -
    mnist/train_mnist.py:284,293,380 unpack exactly two values.  `savedir` is accepted and unused, as in the reference."""
    con = sample(test_images).to(device)
    traj, _ = _eval_common(model, (batch_size, *image_shape), con, "con", solver, steps)
    return traj, con
