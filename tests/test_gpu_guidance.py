"""GPU: the U-Net data gradient (mi355_unet_vjp) and the reconstruction-guidance sampler (AD/image_diffusion/sampling.py:136-206)
against vectors produced by the reference's own UNetModel / DDPM / likelihood.loss under torch.autograd / torch.func.vmap(grad)
(tools/make_goldens.py g_unet_vjp, g_recon_guidance) and against the CPU oracle at the CIFAR configuration.

Tolerances: fp32 mode; a gradient is compared relative to its largest entry (rtol 2e-3, atol 5e-4 * max|ref|); samplers as the
other 25-step DDPM samplers (rtol 2e-3, atol 1e-3)."""
import numpy as np
import pytest
import torch

from mi355.synth import rand_uniform, randn, synth_state_dict
from oracle import ddpm_ref, unet_ref
from tests.test_oracle_golden import NoiseLog, cfg_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(cfg, seed, precision="fp32"):
    from image_diffusion.unet import UNetModel, param_shapes

    net = UNetModel(image_size=cfg.image_size, in_channels=cfg.in_channels, model_channels=cfg.model_channels,
                    out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_resolutions,
                    channel_mult=cfg.channel_mult, conv_resample=cfg.conv_resample, num_heads=cfg.num_heads,
                    num_head_channels=cfg.num_head_channels, num_heads_upsample=cfg.num_heads_upsample,
                    use_scale_shift_norm=cfg.use_scale_shift_norm, resblock_updown=cfg.resblock_updown,
                    use_new_attention_order=cfg.use_new_attention_order, precision=precision)
    sd = synth_state_dict(param_shapes(cfg), seed)
    net.load_state_dict(sd)
    return net.to(DEV), sd


def _close_grad(got, ref, rtol=2e-3, rel_atol=5e-4):
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    print(f"   grad max|err| {err:.3e} (max|ref| {scale:.3e})")
    torch.testing.assert_close(got, ref, rtol=rtol, atol=rel_atol * scale)


@pytest.mark.parametrize("name", ["tiny_in1", "tiny_in3", "tiny_film_updown_neworder", "tiny_noconvresample", "mnist", "cifar", "flowers_in3"])
def test_unet_vjp_vs_reference_autograd(golden, name):
    """(d out / d x)^T g of the differentiable plan against the reference UNetModel under torch.autograd: plain / FiLM / up-down
    ResBlocks, conv and pool resampling, both attention orders, skip concats, 1x1 and 3x3 skips, stride-2 and nearest-x2 convs."""
    g = golden("unet_vjp")
    cfg = cfg_from_json(g.json(f"{name}/config"))
    net, _ = _build(cfg, int(g[f"{name}/seed"]))
    eng = net.engine(DEV, differentiable=True)
    x, t, cot = g.t(f"{name}/x").to(DEV), g.t(f"{name}/t").to(DEV), g.t(f"{name}/g").to(DEV)
    y = eng.forward(x, t)
    torch.testing.assert_close(y.cpu(), g.t(f"{name}/y"), rtol=2e-4, atol=5e-5)   # the differentiable plan's forward is the same network
    gx = eng.vjp(cot, x_channels=cfg.in_channels)
    _close_grad(gx.cpu(), g.t(f"{name}/gx"))
    # linearity in the cotangent and repeatability (the backward pass reads, never destroys, the kept activations)
    gx2 = eng.vjp((2.0 * cot).contiguous(), x_channels=cfg.in_channels)
    torch.testing.assert_close(gx2, 2.0 * gx, rtol=1e-4, atol=1e-5 * float(gx.abs().max()))


def test_unet_vjp_cifar_batch_vs_oracle():
    """A larger batch of the CIFAR net (so the big-tile / persistent conv kernels run the data-gradient convs too): oracle autograd
    on two of the images; bf16 mode bounded."""
    kw = dict(image_size=32, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(2,),
              channel_mult=(1, 2, 2, 2), num_heads=4, num_head_channels=64)
    cfg = unet_ref.UNetConfig(32, 3, 128, 3, 2, (2,), channel_mult=(1, 2, 2, 2), num_heads=4, num_head_channels=64)
    net, sd = _build(cfg, 1234)
    B, pick = 64, [0, 63]
    x, t, cot = randn(31, B, 3, 32, 32), torch.linspace(0.05, 0.95, B), randn(32, B, 3, 32, 32)
    eng = net.engine(DEV, differentiable=True)
    eng.forward(x.to(DEV), t.to(DEV))
    gx = eng.vjp(cot.to(DEV)).cpu()
    xr = x[pick].clone().requires_grad_()
    yr = unet_ref.unet_forward_diff(sd, cfg, xr, t[pick])
    (ref,) = torch.autograd.grad((yr * cot[pick]).sum(), xr)
    _close_grad(gx[pick], ref)
    net.set_precision("bf16")
    eng = net.engine(DEV, differentiable=True)
    eng.forward(x.to(DEV), t.to(DEV))
    g16 = eng.vjp(cot.to(DEV)).cpu()
    rel = float((g16[pick] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"   bf16 vjp relative rms error {rel:.3e}")
    assert torch.isfinite(g16).all() and rel < 0.05


def test_guidance_gradient_probes(golden):
    """x_grad = vmap(grad(constraint))(xi, i, y) at fixed points: seed kernel (loss, clip, predict_start chain rule) + U-Net VJP."""
    from image_diffusion.sde_diffusion import DDPM
    from mi355.ops import default_ops as ops

    g = golden("recon_guidance_tiny")
    Ns = int(g["Ns"])
    cfg = unet_ref.UNetConfig(16, 1, 32, 1, 1, (2,), channel_mult=(1, 2), num_heads=2)
    net, _ = _build(cfg, int(g["net_seed"]))
    eng = net.engine(DEV, differentiable=True)
    T = DDPM(Ns).host_tables()
    for lname, mode in (("paint", 0), ("hyper", 1)):
        cond = g.t(f"probe/{lname}/cond").to(DEV)
        for i in (18, 12, 3):
            xi = g.t(f"probe/{lname}/i{i}/xi").to(DEV)
            eps = eng.forward(xi, torch.full((xi.shape[0],), i / Ns, device=DEV))
            g_eps, g_x = ops.guidance_seed(xi, eps, cond, float(T["sqrt_recip_alphas_cumprod"][i]), float(T["sqrt_recipm1_alphas_cumprod"][i]),
                                           mode, -2.0)
            x_grad = (g_x + eng.vjp(g_eps)).cpu()
            _close_grad(x_grad, g.t(f"probe/{lname}/i{i}/grad"))


def test_reconstruction_guidance_sampler_golden(golden):
    """get_conditional_sample_fn[ReconstructionGuidance] against the reference-driven golden: both update rules, start_fraction,
    a corrector step, Painting.loss and HyperResolution.loss."""
    from image_diffusion import sampling
    from image_diffusion.conditioning import ReconstructionGuidance
    from image_diffusion.likelihoods import HyperResolution, InPainting
    from image_diffusion.sde_diffusion import DDPM

    g = golden("recon_guidance_tiny")
    Ns = int(g["Ns"])
    ddpm = DDPM(Ns)
    cfg = unet_ref.UNetConfig(16, 1, 32, 1, 1, (2,), channel_mult=(1, 2), num_heads=2)
    net, _ = _build(cfg, int(g["net_seed"]))
    eps = sampling.make_eps_model(net, ddpm)
    for tag in ("paint_before", "paint_after", "paint_half_corr1", "hyper_before"):
        lik = InPainting(6, -2) if str(g[f"{tag}/loss"]) == "painting" else HyperResolution(4, 4)
        cond = ReconstructionGuidance(float(g[f"{tag}/gamma"]), float(g[f"{tag}/start_fraction"]), str(g[f"{tag}/rule"]),
                                      int(g[f"{tag}/n_corrector"]), 0.1)
        shape = tuple(g[f"{tag}/xT"].shape)
        base, k = int(g[f"{tag}/noise_base"]), int(g[f"{tag}/draws"])
        with sampling.injected_noise([randn(base + j, *shape) for j in range(k)]):
            x0 = sampling.get_conditional_sample_fn(eps, ddpm, cond, lik)(g.t(f"{tag}/xT").to(DEV), g.t(f"{tag}/cond").to(DEV))
        err = float((x0.cpu() - g.t(f"{tag}/x0")).abs().max())
        print(f"   {tag}: max|err| {err:.3e}")
        torch.testing.assert_close(x0.cpu(), g.t(f"{tag}/x0"), rtol=2e-3, atol=1e-3)
    # an arbitrary callable has no backward pass on this backend: loud error, no fallback
    with pytest.raises(NotImplementedError):
        sampling.get_conditional_sample_fn(lambda xi, i: net(xi, 1.0 * i / Ns), ddpm, ReconstructionGuidance(1.0, 1.0, "before", 0, 0.1),
                                           InPainting(6, -2))
