"""CPU: the oracle (oracle/*.py) against vectors produced by the reference's own modules
(tools/make_goldens.py -> tests/golden/*.npz).  This is what pins the oracle."""
import math

import numpy as np
import pytest
import torch

from mi355.synth import randn, synth_state_dict
from oracle import cfm_ref, ddpm_ref, unet_ref
from oracle.unet_ref import UNetConfig

TOL = dict(rtol=1e-5, atol=1e-5)


def cfg_from_json(c):
    return UNetConfig(
        image_size=c["image_size"], in_channels=c["in_channels"], model_channels=c["model_channels"],
        out_channels=c["out_channels"], num_res_blocks=c["num_res_blocks"],
        attention_resolutions=tuple(c["attention_resolutions"]), channel_mult=tuple(c.get("channel_mult", (1, 2, 4, 8))),
        conv_resample=c.get("conv_resample", True), num_heads=c.get("num_heads", 1),
        num_head_channels=c.get("num_head_channels", -1), num_heads_upsample=c.get("num_heads_upsample", -1),
        use_scale_shift_norm=c.get("use_scale_shift_norm", False), resblock_updown=c.get("resblock_updown", False),
        use_new_attention_order=c.get("use_new_attention_order", False))


def test_timestep_embedding(golden):
    g = golden("timestep_embedding")
    for dim in (32, 128, 33):
        got = unet_ref.timestep_embedding(g.t("t"), dim)
        torch.testing.assert_close(got, g.t(f"dim{dim}"), rtol=0, atol=0)


@pytest.mark.parametrize("Ns", [21, 25, 50, 100, 1000])
def test_ddpm_tables(golden, Ns):
    g = golden("ddpm_tables")
    t = ddpm_ref.ddpm_tables(Ns)
    assert list(g.json("buffer_order")) == list(ddpm_ref.TABLE_NAMES)
    for n in ddpm_ref.TABLE_NAMES:
        torch.testing.assert_close(t[n], g.t(f"Ns{Ns}/{n}"), rtol=0, atol=0, equal_nan=True)


@pytest.mark.parametrize("Ns", [19, 20])
def test_ddpm_tables_nonfinite_quirk(golden, Ns):
    """DDPM(Ns<=20) is non-finite in the reference (SURVEY finding 4): reproduced, not repaired."""
    g = golden("ddpm_tables")
    t = ddpm_ref.ddpm_tables(Ns)
    for n in ddpm_ref.TABLE_NAMES:
        assert np.array_equal(torch.isfinite(t[n]).numpy(), g[f"Ns{Ns}/{n}/isfinite"]), n


def test_groupnorm(golden):
    g = golden("groupnorm")
    for i in range(4):
        y = unet_ref.group_norm32(g.t(f"case{i}/x"), g.t(f"case{i}/w"), g.t(f"case{i}/b"))
        torch.testing.assert_close(y, g.t(f"case{i}/y"), **TOL)


def _resblock_sd(cin, cout, kw, seed, emb=128):
    film = kw.get("use_scale_shift_norm", False)
    shapes = {
        "in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,),
        "in_layers.2.weight": (cout, cin, 3, 3), "in_layers.2.bias": (cout,),
        "emb_layers.1.weight": ((2 if film else 1) * cout, emb), "emb_layers.1.bias": ((2 if film else 1) * cout,),
        "out_layers.0.weight": (cout,), "out_layers.0.bias": (cout,),
        "out_layers.3.weight": (cout, cout, 3, 3), "out_layers.3.bias": (cout,),
    }
    if cout != cin:
        k = 3 if kw.get("use_conv") else 1
        shapes["skip_connection.weight"] = (cout, cin, k, k)
        shapes["skip_connection.bias"] = (cout,)
    return {"rb." + k: v for k, v in synth_state_dict(shapes, seed).items()}


def test_resblock_variants(golden):
    g = golden("resblock")
    for name, (cin, cout, kw) in g.json("variants").items():
        sd = _resblock_sd(cin, cout, kw, int(g[f"{name}/seed"]))
        y = unet_ref.res_block(sd, "rb", g.t(f"{name}/x"), g.t(f"{name}/emb"), cin, cout, kw.get("up", False),
                               kw.get("down", False), kw.get("use_scale_shift_norm", False))
        torch.testing.assert_close(y, g.t(f"{name}/y"), **TOL)


def test_attention(golden):
    g = golden("attention")
    for c in g.json("cases"):
        C = c["C"]
        shapes = {"norm.weight": (C,), "norm.bias": (C,), "qkv.weight": (3 * C, C, 1), "qkv.bias": (3 * C,),
                  "proj_out.weight": (C, C, 1), "proj_out.bias": (C,)}
        sd = {"a." + k: v for k, v in synth_state_dict(shapes, c["seed"]).items()}
        y = unet_ref.attention_block(sd, "a", g.t(c["name"] + "/x"), c["heads"], c["new_order"])
        torch.testing.assert_close(y, g.t(c["name"] + "/y"), **TOL)
    torch.testing.assert_close(unet_ref.qkv_attention(g.t("core/qkv"), 2, False), g.t("core/legacy"), **TOL)
    torch.testing.assert_close(unet_ref.qkv_attention(g.t("core/qkv"), 2, True), g.t("core/new"), **TOL)


def test_updown(golden):
    g = golden("updown")
    x = g.t("x")
    cfg = UNetConfig(8, 32, 32, 32, 1, ())
    sd = {"u.0.conv." + k: v for k, v in synth_state_dict({"weight": (32, 32, 3, 3), "bias": (32,)}, 801).items()}
    torch.testing.assert_close(unet_ref._run_layers(sd, cfg, "u", [("up", 32, True)], x, None), g.t("up_conv"), **TOL)
    torch.testing.assert_close(unet_ref._run_layers({}, cfg, "u", [("up", 32, False)], x, None), g.t("up_nearest"), **TOL)
    sd = {"d.0.op." + k: v for k, v in synth_state_dict({"weight": (32, 32, 3, 3), "bias": (32,)}, 802).items()}
    torch.testing.assert_close(unet_ref._run_layers(sd, cfg, "d", [("down", 32, True)], x, None), g.t("down_conv"), **TOL)
    torch.testing.assert_close(unet_ref._run_layers({}, cfg, "d", [("down", 32, False)], x, None), g.t("down_pool"), **TOL)


UNETS = ["tiny_in1", "tiny_in2", "tiny_in3", "tiny_in6", "tiny_film_updown_neworder", "tiny_noconvresample", "mnist",
         "cifar", "cifar_in6", "flowers_in6", "px128_in6"]


def unet_shapes(cfg):
    """Parameter names/shapes in reference state-dict order, from the product's own module builder."""
    from image_diffusion.unet import param_shapes

    return param_shapes(cfg)


@pytest.mark.parametrize("name", UNETS)
def test_unet_forward(golden, name):
    g = golden("unet_" + name)
    cfg = cfg_from_json(g.json("config"))
    from image_diffusion.unet import param_shapes

    shapes = param_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"])
    sd = synth_state_dict(shapes, int(g["seed"]))
    y = unet_ref.unet_forward(sd, cfg, g.t("x"), g.t("t"))
    torch.testing.assert_close(y, g.t("y"), rtol=2e-4, atol=2e-5)


def _tiny(in_ch, out_ch, seed):
    from image_diffusion.unet import param_shapes

    cfg = UNetConfig(16, in_ch, 32, out_ch, 1, (2,), channel_mult=(1, 2), num_heads=2)
    return cfg, synth_state_dict(param_shapes(cfg), seed)


def test_euler(golden):
    g = golden("euler_tiny")
    cfg, sd = _tiny(3, 3, int(g["seed"]))
    traj = cfm_ref.euler_trajectory(unet_ref.model_fn(sd, cfg), g.t("x0"), torch.linspace(0, 1, 6))
    torch.testing.assert_close(traj, g.t("traj"), rtol=1e-4, atol=1e-5)
    assert (cfm_ref.to_uint8(traj[-1]).int() - g.t("u8").int()).abs().max() <= 1
    torch.testing.assert_close(cfm_ref.to_unit_range(traj[-1]), g.t("unit"), rtol=1e-4, atol=1e-5)


def test_uint8_edge_cases():
    """cifar10/compute_fid.py:87 semantics: truncation, saturation at both ends."""
    x = torch.tensor([-1.0, 1.0, -1.004, 1.004, 0.0, 0.999, -0.5, 0.00392])
    assert cfm_ref.to_uint8(x).tolist() == [0, 255, 0, 255, 128, 255, 64, 128]


def test_ddpm_single_steps(golden):
    g = golden("ddpm_steps")
    d = ddpm_ref.DDPMRef(int(g["Ns"]))
    x, eps, z = g.t("x"), g.t("eps"), g.t("z")
    for i in (0, 1, 12, 24):
        x0 = d.predict_start_from_noise(x, i, eps)
        torch.testing.assert_close(x0, g.t(f"i{i}/x0"), **TOL)
        x0c = x0.clip(-1, 1)
        torch.testing.assert_close(d.q_posterior_mean(x0c, x, i), g.t(f"i{i}/mean"), **TOL)
        nxt = ddpm_ref._ancestral(d, x0c, x, i, lambda s: z)
        torch.testing.assert_close(nxt, g.t(f"i{i}/next"), **TOL)
        torch.testing.assert_close(d.score_from_x0(x0c, i), g.t(f"i{i}/score"), **TOL)
        torch.testing.assert_close(d.q_sample(x, i, z), g.t(f"i{i}/q_sample"), **TOL)


class NoiseLog:
    def __init__(self, base):
        self.base, self.k = base, 0

    def __call__(self, shape):
        z = randn(self.base + self.k, *shape)
        self.k += 1
        return z


STOL = dict(rtol=1e-3, atol=2e-4)  # 25 sequential U-Net calls, fp32 re-association


def test_samplers(golden):
    g = golden("samplers_tiny")
    Ns = int(g["Ns"])
    cfg1, sd1 = _tiny(1, 1, 1001)
    cfg2, sd2 = _tiny(2, 1, 1002)
    net1 = lambda x, t: unet_ref.unet_forward(sd1, cfg1, x, t)
    net2 = lambda x, t: unet_ref.unet_forward(sd2, cfg2, x, t)
    eps1, eps2 = ddpm_ref.make_eps_model(net1, Ns), ddpm_ref.make_eps_model(net2, Ns)

    n = NoiseLog(int(g["prior/noise_base"]))
    x0 = ddpm_ref.prior_sample(eps1, Ns, g.t("prior/xT"), n)
    assert n.k == int(g["prior/draws"]) == Ns - 1
    torch.testing.assert_close(x0, g.t("prior/x0"), **STOL)

    for tag, nc in (("amortized", 0), ("amortized_corr1", 1)):
        n = NoiseLog(int(g[f"{tag}/noise_base"]))
        x0 = ddpm_ref.amortized_sample(eps2, Ns, g.t(f"{tag}/xT"), g.t(f"{tag}/cond"), n, n_corrector=nc, delta=0.1)
        assert n.k == int(g[f"{tag}/draws"])
        torch.testing.assert_close(x0, g.t(f"{tag}/x0"), **STOL)

    n = NoiseLog(int(g["amortized_prior/noise_base"]))
    x0 = ddpm_ref.prior_sample(eps2, Ns, g.t("amortized_prior/xT"), n, amortized=True, none_value=-2.0)
    torch.testing.assert_close(x0, g.t("amortized_prior/x0"), **STOL)

    for tag in ("replacement_noise", "replacement_clean", "replacement_half"):
        n = NoiseLog(int(g[f"{tag}/noise_base"]))
        x0 = ddpm_ref.replacement_sample(eps1, Ns, g.t(f"{tag}/xT"), g.t(f"{tag}/cond"), n,
                                         start_fraction=float(g[f"{tag}/start_fraction"]),
                                         noise_condition=bool(g[f"{tag}/noisy"]))
        assert n.k == int(g[f"{tag}/draws"])
        torch.testing.assert_close(x0, g.t(f"{tag}/x0"), **STOL)


def test_ns20_is_nan(golden):
    g = golden("samplers_tiny")
    assert bool(g["ns20/x0_all_nan_or_inf"])
    cfg1, sd1 = _tiny(1, 1, 1001)
    eps1 = ddpm_ref.make_eps_model(lambda x, t: unet_ref.unet_forward(sd1, cfg1, x, t), 20)
    x0 = ddpm_ref.prior_sample(eps1, 20, randn(1, 1, 1, 16, 16), lambda s: torch.zeros(s))
    assert torch.isnan(x0).all()


def test_likelihoods(golden):
    g = golden("likelihoods")
    img = g.t("img")
    for name, fn in (("inpainting", cfm_ref.inpainting_condition), ("outpainting", cfm_ref.outpainting_condition)):
        hw, patch = g[f"{name}/hw"], int(g[f"{name}/patch"])
        cond = torch.cat([fn(img[k:k + 1], int(hw[k][0]), int(hw[k][1]), patch) for k in range(img.shape[0])])
        torch.testing.assert_close(cond, g.t(f"{name}/cond"), rtol=0, atol=0)
        torch.testing.assert_close(cfm_ref.painting_loss(randn(5002, 3, 3, 32, 32), cond), g.t(f"{name}/loss"), **TOL)
    torch.testing.assert_close(cfm_ref.hyperresolution_condition(g.t("hyper/img"), 16, 16), g.t("hyper/cond"), **TOL)


def _vjp_names(golden):
    return list(golden("unet_vjp").json("names"))


def test_oracle_unet_vjp(golden):
    """The oracle's functional U-Net under torch.autograd against the reference's UNetModel under torch.autograd:
    (d out / d x)^T g - the gradient `vmap(grad(constraint))` needs (sampling.py:154-163)."""
    from image_diffusion.unet import param_shapes

    g = golden("unet_vjp")
    for name in _vjp_names(golden):
        cfg = cfg_from_json(g.json(f"{name}/config"))
        sd = synth_state_dict(param_shapes(cfg), int(g[f"{name}/seed"]))
        x = g.t(f"{name}/x").clone().requires_grad_()
        y = unet_ref.unet_forward_diff(sd, cfg, x, g.t(f"{name}/t"))
        (gx,) = torch.autograd.grad((y * g.t(f"{name}/g")).sum(), x)
        torch.testing.assert_close(y.detach(), g.t(f"{name}/y"), rtol=2e-4, atol=5e-5)
        ref = g.t(f"{name}/gx")
        torch.testing.assert_close(gx, ref, rtol=1e-3, atol=2e-4 * float(ref.abs().max()))


def test_oracle_recon_guidance(golden):
    """oracle/ddpm_ref.recon_guidance_sample (autograd of the summed per-sample losses) against the golden driven by the
    reference's real UNetModel / DDPM / likelihood.loss with its own torch.func.vmap(grad(constraint)) (sampling.py:136-206)."""
    from image_diffusion.unet import param_shapes

    g = golden("recon_guidance_tiny")
    Ns = int(g["Ns"])
    cfg = UNetConfig(16, 1, 32, 1, 1, (2,), channel_mult=(1, 2), num_heads=2)
    sd = synth_state_dict(param_shapes(cfg), int(g["net_seed"]))
    eps = ddpm_ref.make_eps_model(lambda x, t: unet_ref.unet_forward_diff(sd, cfg, x, t), Ns)   # autograd-transparent
    # the constraint gradient at fixed points (no sampler dynamics): autograd of the summed per-sample losses == vmap(grad)
    d = ddpm_ref.DDPMRef(Ns)
    x0_model = ddpm_ref._x0_model(eps, d, False, None)
    for lname, lossf in (("paint", lambda a, b: ddpm_ref.painting_loss(a, b, -2.0)), ("hyper", ddpm_ref.hyperres_loss)):
        for i in (18, 12, 3):
            xr = g.t(f"probe/{lname}/i{i}/xi").clone().requires_grad_()
            (gr,) = torch.autograd.grad(lossf(x0_model(xr, i), g.t(f"probe/{lname}/cond")).sum(), xr)
            ref = g.t(f"probe/{lname}/i{i}/grad")
            torch.testing.assert_close(gr, ref, rtol=1e-3, atol=2e-4 * float(ref.abs().max()))
    for tag in ("paint_before", "paint_after", "paint_half_corr1", "hyper_before"):
        noise = NoiseLog(int(g[f"{tag}/noise_base"]))
        x0 = ddpm_ref.recon_guidance_sample(eps, Ns, g.t(f"{tag}/xT"), g.t(f"{tag}/cond"), lambda shape: noise(shape),
                                            gamma=float(g[f"{tag}/gamma"]), start_fraction=float(g[f"{tag}/start_fraction"]),
                                            update_rule=str(g[f"{tag}/rule"]), n_corrector=int(g[f"{tag}/n_corrector"]), delta=0.1,
                                            loss=str(g[f"{tag}/loss"]))
        assert noise.k == int(g[f"{tag}/draws"])
        torch.testing.assert_close(x0, g.t(f"{tag}/x0"), rtol=2e-3, atol=1e-3)
