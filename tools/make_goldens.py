#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

The reference tree (/root/reference, read-only) cannot travel to the GPU box, so this script is
run once here and its outputs (inputs + expected outputs, a few hundred KB) are committed.

What is imported from the reference, unchanged and without stand-ins for missing libraries:
  AD/image_diffusion/{nn, unet, sde_diffusion, conditioning, likelihoods}.py
via a bare package object (so AD/image_diffusion/__init__.py, which pulls `plum`, never runs).
`sampling.py` (needs un-vendored `plum`), cifar10/*.py and mnist/*.py (absl/torchcfm/torchdyn/
torchvision) are NOT importable: loops from those files are written out below from their source
text and drive the reference's real `UNetModel` / `DDPM` objects, so every arithmetic step of a
golden vector is the reference's, only the loop order is restated ("parity unpinned" for order).

Weights are never stored: they are re-drawn from `mi355.synth.synth_state_dict(shapes, seed)`.
"""
from __future__ import annotations

import importlib
import json
import math
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd")
REF = "/root/reference/amortised diffusion/image_diffusion"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, PKG)

from mi355.synth import rand_uniform, randn, synth_state_dict  # noqa: E402


def import_reference():
    pkg = types.ModuleType("_ref_image_diffusion")
    pkg.__path__ = [REF]
    sys.modules["_ref_image_diffusion"] = pkg
    mods = {}
    for m in ("nn", "unet", "sde_diffusion", "conditioning", "likelihoods"):
        mods[m] = importlib.import_module("_ref_image_diffusion." + m)
    return types.SimpleNamespace(**mods)


ref = import_reference()
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        if isinstance(v, (dict, list, tuple)) and not isinstance(v, np.ndarray):
            v = np.array(json.dumps(v))
        conv[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name:34s} {os.path.getsize(path) / 1024:8.1f} KB")


def load_synth(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(synth_state_dict(shapes, seed))
    module.eval()
    return module


# ---------------------------------------------------------------------------------------------
def g_timestep_embedding():
    t = torch.tensor([0.0, 1e-5, 0.02, 0.5, 0.98, 1.0, 7.0, 999.0])
    out = {"t": t}
    for dim in (32, 128, 33):
        out[f"dim{dim}"] = ref.nn.timestep_embedding(t, dim)
    save("timestep_embedding", **out)


def g_ddpm_tables():
    out = {}
    names = None
    for Ns in (21, 25, 50, 100, 1000):
        d = ref.sde_diffusion.DDPM(Ns)
        names = [n for n, _ in d.named_buffers()]
        for n, b in d.named_buffers():
            out[f"Ns{Ns}/{n}"] = b
        out[f"Ns{Ns}/ts"] = d.ts
    for Ns in (19, 20):
        d = ref.sde_diffusion.DDPM(Ns)
        for n, b in d.named_buffers():
            out[f"Ns{Ns}/{n}/isfinite"] = torch.isfinite(b)
    out["buffer_order"] = names
    save("ddpm_tables", **out)


def g_groupnorm():
    out = {}
    for idx, shape in enumerate([(2, 32, 8, 8), (2, 64, 5, 5), (2, 128, 16), (3, 96, 4, 4)]):
        gn = ref.nn.normalization(shape[1])
        load_synth(gn, 100 + idx)
        x = randn(200 + idx, *shape) * 1.7 + 0.3
        out[f"case{idx}/x"] = x
        out[f"case{idx}/w"] = gn.weight
        out[f"case{idx}/b"] = gn.bias
        out[f"case{idx}/y"] = gn(x)
    save("groupnorm", **out)


RES_VARIANTS = {
    # name: (channels, out_channels, kwargs)
    "plain": (32, 32, {}),
    "chan_1x1": (32, 64, {}),
    "chan_3x3": (32, 64, {"use_conv": True}),
    "film": (64, 64, {"use_scale_shift_norm": True}),
    "film_chan": (32, 64, {"use_scale_shift_norm": True}),
    "up": (32, 32, {"up": True}),
    "down": (32, 32, {"down": True}),
    "concat_odd_groups": (96, 64, {}),
}


def g_resblock():
    out = {"variants": {k: [v[0], v[1], v[2]] for k, v in RES_VARIANTS.items()}, "emb_channels": 128}
    for idx, (name, (cin, cout, kw)) in enumerate(RES_VARIANTS.items()):
        rb = ref.unet.ResBlock(cin, 128, 0.1, out_channels=cout, **kw)
        load_synth(rb, 300 + idx)
        x = randn(400 + idx, 2, cin, 8, 8)
        emb = randn(500 + idx, 2, 128)
        out[f"{name}/x"], out[f"{name}/emb"], out[f"{name}/y"] = x, emb, rb(x, emb)
        out[f"{name}/seed"] = np.int64(300 + idx)
    save("resblock", **out)


def g_attention():
    out = {}
    idx = 0
    cases = []
    for new_order in (False, True):
        for heads in (1, 2):
            for hw in (4, 7):
                ab = ref.unet.AttentionBlock(64, num_heads=heads, use_new_attention_order=new_order)
                load_synth(ab, 600 + idx)
                x = randn(700 + idx, 2, 64, hw, hw)
                name = f"case{idx}"
                out[f"{name}/x"], out[f"{name}/y"] = x, ab(x)
                cases.append({"name": name, "new_order": new_order, "heads": heads, "hw": hw, "seed": 600 + idx, "C": 64})
                idx += 1
    # raw QKV attention cores (no projections): legacy vs new order on the same tensor
    qkv = randn(799, 2, 3 * 2 * 32, 20)
    out["core/qkv"] = qkv
    out["core/legacy"] = ref.unet.QKVAttentionLegacy(2)(qkv)
    out["core/new"] = ref.unet.QKVAttention(2)(qkv)
    out["cases"] = cases
    save("attention", **out)


def g_updown():
    out = {}
    x = randn(800, 2, 32, 8, 8)
    out["x"] = x
    up = load_synth(ref.unet.Upsample(32, True), 801)
    out["up_conv"] = up(x)
    out["up_nearest"] = ref.unet.Upsample(32, False)(x)
    dn = load_synth(ref.unet.Downsample(32, True), 802)
    out["down_conv"] = dn(x)
    out["down_pool"] = ref.unet.Downsample(32, False)(x)
    save("updown", **out)


def make_unet(**kw):
    return ref.unet.UNetModel(**kw)


TINY = dict(image_size=16, model_channels=32, num_res_blocks=1, attention_resolutions=(2,), dropout=0.1,
            channel_mult=(1, 2), num_heads=2)

UNET_CASES = {
    # name: (ctor kwargs, batch, seed)
    "tiny_in1": (dict(TINY, in_channels=1, out_channels=1), 2, 1001),
    "tiny_in2": (dict(TINY, in_channels=2, out_channels=1), 2, 1002),
    "tiny_in3": (dict(TINY, in_channels=3, out_channels=3), 2, 1003),
    "tiny_in6": (dict(TINY, in_channels=6, out_channels=3), 2, 1004),
    "tiny_film_updown_neworder": (dict(TINY, in_channels=3, out_channels=3, use_scale_shift_norm=True,
                                       resblock_updown=True, use_new_attention_order=True,
                                       num_head_channels=32), 2, 1005),
    "tiny_noconvresample": (dict(TINY, in_channels=3, out_channels=3, conv_resample=False), 2, 1006),
    "mnist": (dict(image_size=28, in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                   attention_resolutions=(1,), channel_mult=(1, 2, 2), resblock_updown=True), 2, 1010),
    "cifar": (dict(image_size=32, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2,
                   attention_resolutions=(2,), dropout=0.1, channel_mult=(1, 2, 2, 2), num_heads=4,
                   num_head_channels=64), 2, 1234),
    "cifar_in6": (dict(image_size=32, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=2,
                       attention_resolutions=(2,), dropout=0.1, channel_mult=(1, 2, 2, 2), num_heads=4,
                       num_head_channels=64), 1, 1235),
    "flowers_in6": (dict(image_size=64, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=1,
                         attention_resolutions=(4,), channel_mult=(1, 2, 3, 4), num_heads=4, num_head_channels=64,
                         use_scale_shift_norm=True, resblock_updown=True), 1, 1236),
    # BASELINE cfg 5 geometry (128 px, create_model's 128-px channel_mult (unet.py:68-69), attention at 32/16/8 -> T = 1024/256/64,
    # in = 6) at a quarter of the width so the fixture and the CPU oracle stay small
    "px128_in6": (dict(image_size=128, in_channels=6, model_channels=32, out_channels=3, num_res_blocks=1,
                       attention_resolutions=(4, 8, 16), channel_mult=(1, 1, 2, 3, 4), num_heads=4, num_head_channels=32), 1, 1237),
}


def g_unets():
    for name, (kw, B, seed) in UNET_CASES.items():
        if os.environ.get("ONLY_UNET") and name != os.environ["ONLY_UNET"]:
            continue
        net = load_synth(make_unet(**kw), seed)
        x = randn(seed + 50000, B, kw["in_channels"], kw["image_size"], kw["image_size"])
        t = torch.tensor([0.37, 0.91, 0.0, 1.0][:B])
        y = net(x, t)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        save(f"unet_{name}", x=x, t=t, y=y, config=cfg, seed=np.int64(seed),
             n_params=np.int64(sum(p.numel() for p in net.parameters())))


def tiny_net(in_ch, out_ch, seed):
    return load_synth(make_unet(**dict(TINY, in_channels=in_ch, out_channels=out_ch)), seed)


def g_euler():
    """5-step Euler (loop restated from cifar10/compute_fid.py:76-79 + torchdyn semantics) around the
    reference's real UNetModel, torchcfm call convention model(t, x) = network(x, t.repeat(B))."""
    net = tiny_net(3, 3, 1003)
    x = randn(2001, 2, 3, 16, 16)
    t_span = torch.linspace(0, 1, 6)
    traj = [x]
    for k in range(5):
        t = t_span[k]
        v = net(x, t.repeat(x.shape[0]))
        x = x + (t_span[k + 1] - t) * v
        traj.append(x)
    traj = torch.stack(traj)
    save("euler_tiny", x0=traj[0], traj=traj, u8=(traj[-1] * 127.5 + 128).clip(0, 255).to(torch.uint8),
         unit=traj[-1].clip(-1, 1) / 2 + 0.5, seed=np.int64(1003), steps=np.int64(5))


class NoiseLog:
    """Deterministic injected noise: draw k is mi355.synth.randn(base + k, shape)."""

    def __init__(self, base):
        self.base, self.k = base, 0

    def __call__(self, like):
        z = randn(self.base + self.k, *like.shape)
        self.k += 1
        return z


def g_ddpm_steps():
    """Single-step arithmetic through the reference's DDPM methods (sde_diffusion.py:214-244)."""
    Ns = 25
    d = ref.sde_diffusion.DDPM(Ns)
    x = randn(3001, 2, 3, 8, 8)
    eps = randn(3002, 2, 3, 8, 8)
    z = randn(3003, 2, 3, 8, 8)
    out = {"x": x, "eps": eps, "z": z, "Ns": np.int64(Ns)}
    for i in (0, 1, 12, 24):
        bt = torch.full((2,), i, dtype=torch.long)
        x0 = d.predict_start_from_noise(x, bt, eps)
        x0c = torch.clip(x0, -1, 1)
        mean, var, logvar = d.q_posterior(x0c, x, bt)
        nxt = mean + (0.5 * logvar).exp() * (z if i > 0 else 0.0)
        out[f"i{i}/x0"], out[f"i{i}/mean"], out[f"i{i}/next"] = x0, mean, nxt
        out[f"i{i}/score"] = d.score_from_x0(x0c, bt)
        orig = torch.randn_like
        torch.randn_like = lambda t, **kw: z  # q_sample draws its own noise (:240): inject z
        try:
            out[f"i{i}/q_sample"] = d.q_sample(x, bt)[0]
        finally:
            torch.randn_like = orig
    save("ddpm_steps", **out)


def g_samplers():
    """Reverse samplers at Ns=25 on the tiny net.  Loop order restated from sampling.py (50-75,
    80-133, 209-260); every operation inside a step is the reference's DDPM method / UNetModel."""
    Ns = 25
    d = ref.sde_diffusion.DDPM(Ns)
    B = 2

    def x0_model(net, xi, i, cond=None, amortized=False, none_like=None):
        bt = torch.full((xi.shape[0],), i, dtype=torch.long)
        if amortized:
            if cond is None:
                cond = none_like(xi)
            inp = torch.concat((xi, cond), axis=-3)
        else:
            inp = xi
        eps = net(inp, 1.0 * bt / Ns)
        return torch.clip(d.predict_start_from_noise(xi, bt, eps), -1, 1)

    def step(xi, x0_pred, i, noise):
        bt = torch.full((xi.shape[0],), i, dtype=torch.long)
        mean, var, logvar, _ = d.p_mean_variance(x0_pred, x=xi, i=bt)
        z = noise(xi) if i > 0 else 0.0
        return mean + (0.5 * logvar).exp() * z

    def corrector(net, xi, i, delta, noise, **kw):
        bt = torch.full((xi.shape[0],), i, dtype=torch.long)
        score = d.score_from_x0(x0_model(net, xi, i, **kw), bt)
        dt = (d.tmax - d.tmin) / d.Ns
        return xi + 0.5 * dt * delta * score + math.sqrt(dt * delta) * noise(xi)

    pad = ref.likelihoods.InPainting(patch_size=6, pad_value=-2)
    out = {"Ns": np.int64(Ns)}

    # prior, unconditional net (Replacement/ReconstructionGuidance x0 model), 1 channel
    net = tiny_net(1, 1, 1001)
    xT = randn(4001, B, 1, 16, 16)
    noise = NoiseLog(410000)
    xi = xT
    for i in reversed(range(Ns)):
        xi = step(xi, x0_model(net, xi, i), i, noise)
    out["prior/xT"], out["prior/x0"], out["prior/noise_base"], out["prior/draws"] = xT, torch.clip(xi, -1, 1), np.int64(410000), np.int64(noise.k)
    out["prior/net_seed"] = np.int64(1001)

    # amortized (in = 2*C) with 1 corrector step, inpainting condition with a fixed patch
    net = tiny_net(2, 1, 1002)
    img = rand_uniform(4002, -1, 1, B, 1, 16, 16)
    cond = img.clone()
    cond[:, :, 5:11, 4:10] = -2.0
    xT = randn(4003, B, 1, 16, 16)
    for tag, ncorr in (("amortized", 0), ("amortized_corr1", 1)):
        noise = NoiseLog(420000 if ncorr == 0 else 430000)
        xi = xT
        for i in reversed(range(Ns)):
            xi = step(xi, x0_model(net, xi, i, cond, amortized=True, none_like=pad.none_like), i, noise)
            for _ in range(ncorr):
                xi = corrector(net, xi, i, 0.1, noise, amortized=True, none_like=pad.none_like)
        out[f"{tag}/xT"], out[f"{tag}/cond"], out[f"{tag}/x0"] = xT, cond, torch.clip(xi, -1, 1)
        out[f"{tag}/noise_base"], out[f"{tag}/draws"] = np.int64(noise.base), np.int64(noise.k)
        out[f"{tag}/net_seed"] = np.int64(1002)

    # amortized prior (condition = none_like) - what get_prior_sample_fn does under Amortized
    noise = NoiseLog(440000)
    xi = xT
    for i in reversed(range(Ns)):
        xi = step(xi, x0_model(net, xi, i, None, amortized=True, none_like=pad.none_like), i, noise)
    out["amortized_prior/xT"], out["amortized_prior/x0"] = xT, torch.clip(xi, -1, 1)
    out["amortized_prior/noise_base"] = np.int64(440000)

    # replacement, noise on / off, start_fraction 1.0 and 0.5
    net = tiny_net(1, 1, 1001)
    for tag, noisy, sf, base in (("replacement_noise", True, 1.0, 450000), ("replacement_clean", False, 1.0, 460000),
                                 ("replacement_half", True, 0.5, 470000)):
        noise = NoiseLog(base)
        xi = xT
        for i in reversed(range(Ns)):
            bt = torch.full((B,), i, dtype=torch.long)
            if i < int(Ns * sf):
                if noisy:
                    orig = torch.randn_like
                    torch.randn_like = lambda t, **kw: noise(t)
                    try:
                        nc, _ = d.q_sample(cond, bt)
                    finally:
                        torch.randn_like = orig
                else:
                    nc = cond
                xi = torch.where(cond == pad.pad_value, xi, nc)
            xi = step(xi, x0_model(net, xi, i), i, noise)
        out[f"{tag}/xT"], out[f"{tag}/cond"], out[f"{tag}/x0"] = xT, cond, torch.clip(xi, -1, 1)
        out[f"{tag}/noise_base"], out[f"{tag}/draws"] = np.int64(base), np.int64(noise.k)
        out[f"{tag}/start_fraction"], out[f"{tag}/noisy"] = np.float64(sf), np.bool_(noisy)

    # Ns = 20 => NaN (SURVEY finding 4): known-answer quirk
    d20 = ref.sde_diffusion.DDPM(20)
    bt = torch.full((B,), 19, dtype=torch.long)
    x0 = d20.predict_start_from_noise(xT, bt, xT)
    out["ns20/x0_all_nan_or_inf"] = np.bool_(bool((~torch.isfinite(x0)).all()))
    save("samplers_tiny", **out)


def g_unet_vjp():
    """Vector-Jacobian products (d out / d x)^T g of the reference's real UNetModel under torch.autograd, the quantity
    `vmap(grad(constraint))` needs (sampling.py:154-163): input x, times t, cotangent g -> expected gradient."""
    out = {}
    cases = {"tiny_in1": 2101, "tiny_in3": 2103, "tiny_film_updown_neworder": 2105, "tiny_noconvresample": 2106, "mnist": 2110, "cifar": 2134,
             "flowers_in3": 2136}
    kws = dict(UNET_CASES)
    kws["flowers_in3"] = (dict(UNET_CASES["flowers_in6"][0], in_channels=3), 1, 1236)
    names = []
    for name, gseed in cases.items():
        kw, B, seed = kws[name]
        net = load_synth(make_unet(**kw), seed)
        x = randn(seed + 50000, B, kw["in_channels"], kw["image_size"], kw["image_size"])
        t = torch.tensor([0.37, 0.91, 0.0, 1.0][:B])
        g = randn(gseed, B, kw["out_channels"], kw["image_size"], kw["image_size"])
        with torch.enable_grad():
            xr = x.clone().requires_grad_()
            y = net(xr, t)
            (gx,) = torch.autograd.grad((y * g).sum(), xr)
        out[f"{name}/x"], out[f"{name}/t"], out[f"{name}/g"], out[f"{name}/gx"], out[f"{name}/y"] = x, t, g, gx, y.detach()
        out[f"{name}/config"] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        out[f"{name}/seed"] = np.int64(seed)
        names.append(name)
    out["names"] = names
    save("unet_vjp", **out)


def g_recon_guidance():
    """Reconstruction-guidance sampler (sampling.py:136-206) around the reference's real UNetModel, DDPM methods and
    Painting.loss / HyperResolution.loss, with the reference's own torch.func.vmap(grad(constraint)) gradient; only the loop is
    restated (sampling.py needs the un-vendored plum).  Both update rules, start_fraction 1.0 and 0.5, one corrector case."""
    from torch.func import grad, vmap

    Ns, B = 25, 2
    d = ref.sde_diffusion.DDPM(Ns)
    net = tiny_net(1, 1, 1001)
    pad = ref.likelihoods.InPainting(patch_size=6, pad_value=-2)
    hyper = ref.likelihoods.HyperResolution(4, 4)
    img = rand_uniform(6002, -1, 1, B, 1, 16, 16)
    cond_paint = img.clone()
    cond_paint[:, :, 5:11, 4:10] = -2.0
    cond_hyper = hyper.sample(img)
    xT = randn(6003, B, 1, 16, 16)

    def x0_model(xi, i):
        bt = i if torch.is_tensor(i) else torch.full((xi.shape[0],), i, dtype=torch.long)
        eps = net(xi, 1.0 * bt / d.Ns)
        return torch.clip(d.predict_start_from_noise(xi, bt, eps), -1, 1)

    out = {"Ns": np.int64(Ns), "net_seed": np.int64(1001)}
    # gamma: the update is gamma * alpha_i (1 - alpha_i) * x_grad with x_grad ~ sqrt_recip_alphas_cumprod[i] * 2 (x0 - y) (up to 2e3 at
    # the first steps): with UNTRAINED (synthetic) weights larger values make the guided chain chaotic (a 1e-7 difference ends
    # anywhere in [-1, 1]), which would pin nothing; these keep it contractive while moving x0 by O(0.1)
    cases = [("paint_before", pad, cond_paint, 0.02, 1.0, "before", 0), ("paint_after", pad, cond_paint, 0.02, 1.0, "after", 0),
             ("paint_half_corr1", pad, cond_paint, 3.0, 0.5, "before", 1), ("hyper_before", hyper, cond_hyper, 1.0, 1.0, "before", 0)]
    # the gradient itself at fixed points (pure function of (xi, i, y)): the chain-rule pieces without any sampler dynamics
    for lname, lik, condition in (("paint", pad, cond_paint), ("hyper", hyper, cond_hyper)):
        for i in (18, 12, 3):
            bt = torch.full((B,), i, dtype=torch.long)
            xi = xT * float(d.sqrt_one_minus_alphas_cumprod[i]) + img * float(d.sqrt_alphas_cumprod[i])   # a plausible x_i
            def constraint(x1, i1, y1, lik=lik):
                x0 = x0_model(x1.unsqueeze(0), i1.unsqueeze(0))
                return lik.loss(x0, y1.unsqueeze(0)).squeeze(0) if lik is pad else lik.loss(x0, y1.unsqueeze(0))
            with torch.enable_grad():
                out[f"probe/{lname}/i{i}/grad"] = vmap(grad(constraint, argnums=0))(xi.clone(), bt, condition).detach()
            out[f"probe/{lname}/i{i}/xi"] = xi
        out[f"probe/{lname}/cond"] = condition
    for k, (tag, lik, condition, gamma, sf, rule, ncorr) in enumerate(cases):
        noise = NoiseLog(610000 + 10000 * k)
        xi = xT.clone()
        first_grad = None
        for i in reversed(range(Ns)):
            bt = torch.full((B,), i, dtype=torch.long)
            x_update = 0.0
            if i < int(Ns * sf):
                def constraint(x1, i1, y1):
                    x0 = x0_model(x1.unsqueeze(0), i1.unsqueeze(0))
                    return lik.loss(x0, y1.unsqueeze(0)).squeeze(0) if lik is pad else lik.loss(x0, y1.unsqueeze(0))
                with torch.enable_grad():
                    x_grad = vmap(grad(constraint, argnums=0))(xi.detach().clone(), bt, condition)
                if first_grad is None:
                    first_grad = x_grad.detach().clone()
                alpha_i = d.alphas[i]
                x_update = -(gamma * alpha_i * (1 - alpha_i)) * x_grad
                if rule == "before":
                    xi = xi + x_update
            x0_pred = x0_model(xi, bt)
            mean, var, logvar, _ = d.p_mean_variance(x0_pred, x=xi, i=bt)
            z = noise(xi) if i > 0 else 0.0
            pred = mean + (0.5 * logvar).exp() * z
            if rule == "after":
                pred = pred + x_update
            xi = pred
            for _ in range(ncorr):
                score = d.score_from_x0(x0_model(xi, bt), bt)
                dt = (d.tmax - d.tmin) / d.Ns
                xi = xi + 0.5 * dt * 0.1 * score + math.sqrt(dt * 0.1) * noise(xi)
        out[f"{tag}/xT"], out[f"{tag}/cond"], out[f"{tag}/x0"], out[f"{tag}/first_grad"] = xT, condition, torch.clip(xi, -1, 1), first_grad
        out[f"{tag}/gamma"], out[f"{tag}/start_fraction"], out[f"{tag}/n_corrector"] = np.float64(gamma), np.float64(sf), np.int64(ncorr)
        out[f"{tag}/noise_base"], out[f"{tag}/draws"], out[f"{tag}/rule"] = np.int64(noise.base), np.int64(noise.k), rule
        out[f"{tag}/loss"] = "painting" if lik is pad else "hyperres"
    save("recon_guidance_tiny", **out)


def g_likelihoods():
    out = {}
    img = rand_uniform(5001, -1, 1, 3, 3, 32, 32)
    out["img"] = img
    for name, cls, patch in (("inpainting", ref.likelihoods.InPainting, 8), ("outpainting", ref.likelihoods.OutPainting, 8)):
        lik = cls(patch_size=patch, pad_value=-2)
        torch.manual_seed(77)
        hw = []
        for _ in range(img.shape[0]):
            h, w = lik.get_random_patch(32)
            hw.append((int(h), int(w)))
        torch.manual_seed(77)
        cond = lik.sample(img)
        out[f"{name}/cond"], out[f"{name}/hw"], out[f"{name}/patch"] = cond, np.array(hw), np.int64(patch)
        out[f"{name}/none_like"] = lik.none_like(img[:1])
        out[f"{name}/loss"] = lik.loss(randn(5002, 3, 3, 32, 32), cond)
    hr = ref.likelihoods.HyperResolution(16, 16)
    img64 = rand_uniform(5003, -1, 1, 2, 3, 64, 64)
    out["hyper/img"], out["hyper/cond"] = img64, hr.sample(img64)
    out["hyper/none_like"] = hr.none_like(img64[:1])
    save("likelihoods", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    gens = [g_timestep_embedding, g_ddpm_tables, g_groupnorm, g_resblock, g_attention, g_updown, g_unets, g_euler,
            g_ddpm_steps, g_samplers, g_likelihoods, g_unet_vjp, g_recon_guidance]
    for g in gens:
        if not only or g.__name__ in only:
            g()


def g_unet_keys():
    """state_dict key order + shapes of the reference UNetModel for the three reference configs."""
    out = {}
    for name in ("cifar", "mnist", "flowers_in6", "tiny_noconvresample"):
        kw, _, _ = UNET_CASES[name]
        sd = make_unet(**kw).state_dict()
        out[name] = [[k, list(v.shape)] for k, v in sd.items()]
    save("unet_keys", keys=out)


if __name__ == "__main__" and ("g_unet_keys" in sys.argv[1:] or len(sys.argv) == 1):
    g_unet_keys()
