"""GPU parity of the single HIP ops (through the C ABI) against the CPU oracle and the reference goldens."""
import math

import pytest
import torch
import torch.nn.functional as F

from mi355 import _lib
from mi355.synth import rand_uniform, randn, synth_state_dict
from oracle import cfm_ref, ddpm_ref, unet_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from mi355.ops import default_ops

    return default_ops


def test_timestep_embedding(ops, golden):
    g = golden("timestep_embedding")
    t = g.t("t").to(DEV)
    for dim in (32, 128, 33):
        got = ops.timestep_embedding(t, dim).cpu()
        # fp32 sin/cos of arguments up to 999: device libm vs torch's differ by a few ulp of the ARGUMENT
        torch.testing.assert_close(got, g.t(f"dim{dim}"), rtol=0, atol=2e-4)
        small = g.t("t") <= 7
        torch.testing.assert_close(got[small], g.t(f"dim{dim}")[small], rtol=0, atol=2e-6)


def test_groupnorm_op(ops, golden):
    g = golden("groupnorm")
    for i in range(4):
        x, w, b = g.t(f"case{i}/x").to(DEV), g.t(f"case{i}/w").to(DEV), g.t(f"case{i}/b").to(DEV)
        y = ops.groupnorm(x.contiguous(), w, b).cpu()
        torch.testing.assert_close(y, g.t(f"case{i}/y"), rtol=1e-4, atol=1e-5)
        ys = ops.groupnorm(x.contiguous(), w, b, silu=True).cpu()
        torch.testing.assert_close(ys, F.silu(g.t(f"case{i}/y")), rtol=1e-4, atol=1e-5)


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, resample
    (2, 32, 8, 8, 64, 3, 1, 0),
    (3, 64, 16, 16, 128, 3, 1, 0),
    (2, 3, 32, 32, 128, 3, 1, 0),     # first conv (channel-padded input)
    (2, 128, 32, 32, 3, 3, 1, 0),     # last conv (NCHW fp32 epilogue)
    (2, 96, 4, 4, 32, 3, 1, 0),       # several images per tile
    (5, 32, 28, 28, 32, 3, 1, 0),     # non power-of-two image (MNIST)
    (2, 64, 14, 14, 64, 3, 1, 0),
    (3, 64, 7, 7, 64, 3, 1, 0),
    (2, 64, 16, 16, 64, 3, 2, 0),     # Downsample conv
    (2, 32, 7, 7, 32, 3, 2, 0),
    (2, 64, 8, 8, 64, 3, 1, 2),       # Upsample: nearest x2 + conv
    (2, 32, 8, 8, 32, 3, 1, 3),       # avg-pool gather
    (2, 128, 16, 16, 384, 1, 1, 0),   # qkv 1x1
    (2, 256, 8, 8, 128, 1, 1, 0),     # skip 1x1
    (1, 32, 64, 64, 64, 3, 1, 0),
    (1, 32, 128, 128, 32, 3, 1, 0),   # one tile row per workgroup
    # bench-sized launches (>= 512 workgroups of 128 pixels x 128 channels): the warp-specialised persistent kernel
    (64, 128, 32, 32, 128, 3, 1, 0),
    (64, 64, 32, 32, 256, 3, 1, 0),   # two channel tiles per pixel tile
    (64, 128, 16, 16, 128, 3, 1, 2),  # nearest x2 gather
    (80, 64, 28, 28, 128, 3, 1, 0),   # ragged tiles, tile count not a multiple of the persistent grid
    (16, 64, 64, 64, 128, 3, 1, 0),   # 4 x 4 tiles per image (BASELINE cfg 4 / 5 image sizes)
    (256, 64, 8, 8, 256, 3, 1, 0),    # bench-sized 8x8 level
    (255, 128, 8, 8, 256, 3, 1, 0),   # odd batch
]


def _conv_ref(x, w, b, k, stride, resample):
    if resample == 2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif resample == 3:
        x = F.avg_pool2d(x, 2, 2)
    return F.conv2d(x, w, b, stride=stride, padding=k // 2)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype,rtol,atol", [(_lib.MI355_F32, 2e-5, 2e-5), (_lib.MI355_BF16, 2e-2, 2e-2), (_lib.MI355_F16, 3e-3, 3e-3)])
def test_conv2d(ops, case, dtype, rtol, atol):
    B, Cin, H, W, Cout, k, stride, resample = case
    seed = hash(case) % 10000
    x = randn(seed, B, Cin, H, W)
    sd = synth_state_dict({"weight": (Cout, Cin, k, k), "bias": (Cout,)}, seed + 1)
    ref = _conv_ref(x, sd["weight"], sd["bias"], k, stride, resample)
    got = ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], stride=stride, resample=resample, dtype=dtype).cpu()
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("dtype,rtol,atol", [(_lib.MI355_F32, 5e-5, 5e-5), (_lib.MI355_BF16, 3e-2, 3e-2), (_lib.MI355_F16, 4e-3, 4e-3)])
def test_gn_silu_conv_fused(ops, silu, dtype, rtol, atol):
    """GroupNorm32 (+SiLU) folded into the conv's staging prologue == ResBlock in_layers (unet.py:283-286)."""
    for (B, C, H, Co) in [(2, 64, 16, 128), (3, 96, 8, 64), (9, 32, 4, 32), (67, 128, 32, 128)]:
        x = randn(C + H, B, C, H, H) * 1.5 + 0.2
        sd = synth_state_dict({"in_layers.0.weight": (C,), "in_layers.0.bias": (C,), "weight": (Co, C, 3, 3), "bias": (Co,)}, C)
        h = unet_ref.group_norm32(x, sd["in_layers.0.weight"], sd["in_layers.0.bias"])
        ref = F.conv2d(F.silu(h) if silu else h, sd["weight"], sd["bias"], padding=1)
        got = ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], gn=(sd["in_layers.0.weight"].to(DEV), sd["in_layers.0.bias"].to(DEV)),
                         gn_silu=silu, dtype=dtype).cpu()
        torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype,rtol,atol", [(_lib.MI355_F32, 5e-5, 5e-5), (_lib.MI355_BF16, 3e-2, 3e-2), (_lib.MI355_F16, 4e-3, 4e-3)])
def test_out_conv_streaming_kernel(ops, dtype, rtol, atol):
    """UNetModel.out = GroupNorm32 -> SiLU -> conv3x3 to <= 4 channels, NCHW fp32 (unet.py:702-706) on the streaming kernel of conv_edge.hip
    (conv_edge bit 0) and on the generic tile kernel it replaces: whole images, ragged 20 x 28 images (partial tiles, zero padding applied
    AFTER the activation), 256 input channels (two fragment groups per pixel in fp32), out_channels 1 .. 4."""
    for (B, C, H, W, Co) in [(5, 128, 32, 32, 3), (3, 128, 20, 28, 3), (2, 256, 16, 16, 1), (2, 128, 64, 64, 4), (300, 128, 8, 8, 2),
                             (70, 64, 32, 32, 3), (1, 64, 16, 16, 3), (2, 128, 128, 128, 3)]:   # 64 channels: the fp32 build's counted path; one tile; many tiles per workgroup
        x = randn(300 + C + H, B, C, H, W) * 1.7 + 0.3
        sd = synth_state_dict({"in_layers.0.weight": (C,), "in_layers.0.bias": (C,), "weight": (Co, C, 3, 3), "bias": (Co,)}, C + Co)
        h = F.silu(unet_ref.group_norm32(x, sd["in_layers.0.weight"], sd["in_layers.0.bias"]))
        ref = F.conv2d(h, sd["weight"], sd["bias"], padding=1)
        gn = (sd["in_layers.0.weight"].to(DEV), sd["in_layers.0.bias"].to(DEV))
        got = ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], gn=gn, gn_silu=True, dtype=dtype, debug=_lib.debug_config(conv_edge=1)).cpu()
        old = ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], gn=gn, gn_silu=True, dtype=dtype, debug=_lib.debug_config(conv_edge=0)).cpu()
        torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)
        torch.testing.assert_close(got, old, rtol=rtol, atol=atol)


def test_first_conv_kernel(ops):
    """UNetModel.input_blocks[0] = conv3x3(in_channels -> model_channels) (unet.py:575) in bf16 mode on conv3x3_in_kernel (conv_edge bit 1:
    contraction over tap x the 8 channels of a pixel's first 16-byte slot) against torch and against the generic tile kernel on the same
    inputs: 3 / 6 / 1 / 8 input channels (x alone, x || condition, MNIST, a full slot), one tile, many tiles, 128-pixel images."""
    for (B, Ci, H, W) in [(5, 3, 32, 32), (3, 6, 32, 32), (2, 1, 16, 16), (2, 8, 48, 16), (2, 6, 128, 128), (70, 3, 64, 64)]:
        x = randn(700 + Ci + H, B, Ci, H, W) * 1.3
        sd = synth_state_dict({"weight": (128, Ci, 3, 3), "bias": (128,)}, 710 + Ci)
        ref = F.conv2d(x, sd["weight"], sd["bias"], padding=1)
        got = ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], dtype=_lib.MI355_BF16, debug=_lib.debug_config(conv_edge=3)).cpu()
        old = ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], dtype=_lib.MI355_BF16, debug=_lib.debug_config(conv_edge=1)).cpu()
        torch.testing.assert_close(got, ref, rtol=3e-2, atol=3e-2)
        torch.testing.assert_close(got, old, rtol=1e-2, atol=1e-2)   # same bf16 operands, same bf16 output rounding: only the fp32 summation order differs
        # tight check of the bf16-only kernel (no fp32 instantiation to hold to 5e-5): against fp32 math on the SAME bf16-rounded operands only the
        # summation order and the final bf16 store remain - within one bf16 ulp of the value element by element, and zero-mean: a misplaced bias or
        # tap of 1e-3 shows up in the per-channel mean error (random roundings average out over >= 512 pixels x B images)
        xb, wb = x.bfloat16().float(), sd["weight"].bfloat16().float()
        refb = F.conv2d(xb, wb, sd["bias"], padding=1)
        err = got - refb
        assert (err.abs() <= refb.abs() * 2.0 ** -7 + 2e-5).all(), float((err.abs() - refb.abs() * 2.0 ** -7).max())
        bias_c = err.mean(dim=(0, 2, 3)).abs().max().item()
        noise = 6.0 * err.std().item() / (B * H * W) ** 0.5        # 6 sigma of a per-channel mean of zero-mean rounding errors (max over 128 channels)
        assert bias_c < noise + 1e-5, (bias_c, noise)


@pytest.mark.parametrize("dtype,tol", [(_lib.MI355_F32, 2e-5), (_lib.MI355_BF16, 2e-2), (_lib.MI355_F16, 3e-3)])
def test_qkv_attention(ops, golden, dtype, tol):
    g = golden("attention")
    qkv = g.t("core/qkv").to(DEV)
    torch.testing.assert_close(ops.qkv_attention(qkv, 2, False, dtype).cpu(), g.t("core/legacy"), rtol=tol, atol=tol)
    torch.testing.assert_close(ops.qkv_attention(qkv, 2, True, dtype).cpu(), g.t("core/new"), rtol=tol, atol=tol)
    shapes = [(2, 4, 64, 256), (3, 1, 32, 784), (2, 1, 64, 49), (2, 4, 64, 16), (1, 2, 128, 100),
              (2, 4, 64, 1024),                      # BASELINE cfg 5: attention at 32x32 (two query blocks per wave, 16 key tiles)
              (1, 3, 32, 300), (2, 1, 96, 256),      # ragged length with two query blocks; torchcfm single head at 96 channels
              (1, 1, 192, 70), (1, 1, 256, 256),     # single-head torchcfm defaults: 32-key tiles
              (2, 1, 384, 256), (2, 1, 512, 64)]     # SuperResModelWrapper(dim=(3,64,64), 128 ch) at 16x16 / 8x8 (train_mnist_hy.py:312-318)
    for (B, heads, ch, T) in shapes:
        q = randn(T + ch, B, 3 * heads * ch, T)
        for new in (False, True):
            ref = unet_ref.qkv_attention(q, heads, new)
            got = ops.qkv_attention(q.to(DEV), heads, new, dtype).cpu()
            torch.testing.assert_close(got, ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype,tol", [(_lib.MI355_F32, 1e-4), (_lib.MI355_BF16, 3e-2), (_lib.MI355_F16, 4e-3)])
def test_attention_softmax_spike(ops, dtype, tol):
    """Online-softmax slow path: a key far above the rest appears in a LATER tile (guide rule 26), so the column's reference offset -
    the start value of its S^T accumulators - has to move and O / l are rescaled; a second case puts every logit far BELOW zero (the
    first tile's offset is its own maximum, whatever the sign: exp2 of the raw logits would underflow to an all-zero row)."""
    B, heads, ch, T = 1, 1, 64, 256
    q = randn(11, B, 3 * ch, T) * 0.5
    q[0, ch:2 * ch, 200] = q[0, 0:ch, 7] * 40.0  # key 200 aligned with query 7: huge logit in the 4th key tile
    ref = unet_ref.qkv_attention(q, heads, False)
    got = ops.qkv_attention(q.to(DEV), heads, False, dtype).cpu()
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol)
    q2 = randn(12, B, 3 * ch, T) * 0.5
    q2[0, 0:ch] = 6.0 + 0.1 * q2[0, 0:ch]            # q ~ +6, k ~ -6: logits ~ -64 * 36 / 8 = -288 for every key
    q2[0, ch:2 * ch] = -6.0 + 0.1 * q2[0, ch:2 * ch]
    ref2 = unet_ref.qkv_attention(q2, heads, False)
    got2 = ops.qkv_attention(q2.to(DEV), heads, False, dtype).cpu()
    assert torch.isfinite(got2).all()
    torch.testing.assert_close(got2, ref2, rtol=max(tol, 2e-3), atol=max(tol, 2e-3))


def test_step_kernels(ops, golden):
    g = golden("ddpm_steps")
    d = ddpm_ref.DDPMRef(int(g["Ns"]))
    T = d.t
    x, eps, z = g.t("x"), g.t("eps"), g.t("z")
    for i in (0, 1, 12, 24):
        xi = x.to(DEV).clone()
        sigma = float((0.5 * T["posterior_log_variance_clipped"][i]).exp())
        ops.ddpm_step_(xi, eps.to(DEV), z.to(DEV) if i > 0 else None, float(T["sqrt_recip_alphas_cumprod"][i]),
                       float(T["sqrt_recipm1_alphas_cumprod"][i]), float(T["posterior_mean_coef1"][i]),
                       float(T["posterior_mean_coef2"][i]), sigma)
        torch.testing.assert_close(xi.cpu(), g.t(f"i{i}/next"), rtol=1e-5, atol=1e-5)
        # replacement: q_sample on the condition where it is not the sentinel
        cond = x.clone()
        cond[:, :, 2:5, 1:6] = -2.0
        xr = eps.to(DEV).clone()
        ops.replace_mask_(xr, cond.to(DEV), z.to(DEV), -2.0, True, float(T["sqrt_alphas_cumprod"][i]),
                          float(T["sqrt_one_minus_alphas_cumprod"][i]))
        ref = torch.where(cond == -2.0, eps, d.q_sample(cond, i, z))
        torch.testing.assert_close(xr.cpu(), ref, rtol=1e-6, atol=1e-6)
        # corrector
        xc = x.to(DEV).clone()
        ops.corrector_step_(xc, eps.to(DEV), z.to(DEV), float(T["sqrt_recip_alphas_cumprod"][i]), float(T["sqrt_recipm1_alphas_cumprod"][i]),
                            float(T["recip_sqrt_m1_alphas_cumprod"][i]), (1.0 - 1e-5) / 25, 0.1)
        refc = ddpm_ref._corrector(d, lambda xx, ii: d.predict_start_from_noise(xx, ii, eps).clip(-1, 1), x.clone(), i, 0.1, lambda s: z)
        torch.testing.assert_close(xc.cpu(), refc, rtol=1e-5, atol=1e-5)


def test_euler_clip_quantize(ops):
    x = randn(5, 3, 3, 17, 19) * 1.3
    v = randn(6, 3, 3, 17, 19)
    got = ops.euler_step_(x.to(DEV).clone(), v.to(DEV), 0.02).cpu()
    torch.testing.assert_close(got, x + 0.02 * v, rtol=0, atol=1e-7)
    assert torch.equal(ops.quantize_u8(x.to(DEV)).cpu(), cfm_ref.to_uint8(x))
    edge = torch.tensor([-1.0, 1.0, -1.004, 1.004, 0.0, 0.999, -0.5, 0.00392, 3.0, -3.0, 0.5])
    assert ops.quantize_u8(edge.to(DEV)).cpu().tolist() == cfm_ref.to_uint8(edge).tolist()
    torch.testing.assert_close(ops.to_unit_range(x.to(DEV)).cpu(), cfm_ref.to_unit_range(x), rtol=0, atol=1e-7)
    xn = x.clone()
    xn[0, 0, 0, :3] = float("nan")
    c = ops.clip_(xn.to(DEV).clone(), -1, 1).cpu()
    assert torch.isnan(c[0, 0, 0, :3]).all()  # torch.clip propagates NaN (DDPM(Ns<=20) quirk depends on it)
    torch.testing.assert_close(c, xn.clip(-1, 1), rtol=0, atol=0, equal_nan=True)


def test_philox_randn_moments(ops):
    z = ops.randn((1 << 20,), DEV, seed=1234, offset=0)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1.0) < 5e-3
    assert abs((z ** 3).mean().item()) < 2e-2 and abs((z ** 4).mean().item() - 3.0) < 5e-2
    z2 = ops.randn((1 << 20,), DEV, seed=1234, offset=0)
    assert torch.equal(z, z2)  # counter-based: reproducible
    z3 = ops.randn((1 << 10,), DEV, seed=1234, offset=1 << 10)
    assert torch.equal(z3, z[1 << 10: 1 << 11])  # offset addresses the same stream


@pytest.mark.gpu
def test_ema_update_matches_reference_expression(ops):
    """utils_cifar.ema (cifar10/utils_cifar.py:47-53): target*decay + source*(1-decay), bit-exact vs the eager fp32 expression."""
    t = randn(11, 3, 1000, 37)
    s = randn(12, 3, 1000, 37)
    for decay in (0.9999, 0.5, 0.0):
        ref = t * decay + s * (1 - decay)
        got = ops.ema_update_(t.clone().to(DEV), s.to(DEV), decay).cpu()
        assert torch.equal(got, ref)


@pytest.mark.parametrize("shape,size", [((3, 3, 64, 64), (16, 16)), ((2, 3, 16, 16), (64, 64)), ((5, 1, 28, 28), (7, 7)), ((5, 1, 7, 7), (28, 28)),
                                        ((2, 2, 13, 9), (31, 20)), ((1, 1, 5, 5), (5, 5))])
def test_resize_bilinear_vs_torch(ops, shape, size):
    """mi355_resize_bilinear = F.interpolate(mode="bilinear", align_corners=False): downsample_images (mnist/utils_mnist_hy.py:18-28),
    HyperResolution._sample (likelihoods.py:119-126), the SuperRes wrapper's up-sampling.  fp32: 1e-6 (same taps, same weights; only
    the contraction of the four products may differ)."""
    x = randn(500 + shape[2], *shape)
    want = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    got = ops.resize_bilinear(x.to(DEV), size).cpu()
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    # linearity (what lets the drifting low-res condition be up-sampled once per solve): up(a x) == a up(x) to rounding
    a = 1.0 + 1.0 / 99
    torch.testing.assert_close(ops.resize_bilinear((x * a).to(DEV), size).cpu(), got * a, rtol=1e-6, atol=1e-6)


def test_likelihood_builders_on_device_match_the_host_loop():
    """Likelihood.sample (likelihoods.py:22-27: one draw per image in a Python loop): on device tensors the draws stay on the host in
    the same order and the tensor work is one launch - bit-identical to the CPU path under the same seed, for all three types."""
    from image_diffusion.likelihoods import HyperResolution, InPainting, OutPainting

    x = rand_uniform(77, -1.0, 1.0, 9, 3, 32, 32)
    for lik in (InPainting(8, -2.0), OutPainting(12, -2.0)):
        torch.manual_seed(3)
        want = lik.sample(x)
        torch.manual_seed(3)
        got = lik.sample(x.to(DEV))
        assert got.is_cuda and torch.equal(got.cpu(), want)
        assert float((want == -2.0).float().mean()) > 0
    h = HyperResolution(16, 16)
    torch.testing.assert_close(h.sample(x.to(DEV)).cpu(), h.sample(x), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(h.loss(x.to(DEV), h.sample(x.to(DEV))).cpu(), h.loss(x, h.sample(x)), rtol=1e-5, atol=1e-7)


def test_forward_with_host_scalar_time_matches_tensor_time():
    """mi355_unet_forward_t (one host-side time for the batch, as the ODE solvers call the field) == mi355_unet_forward with t.repeat(B) (to fp32 rounding)."""
    from image_diffusion.unet import UNetModel, param_shapes

    net = UNetModel(image_size=16, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=(2,),
                    channel_mult=(1, 2), num_heads=2, precision="fp32")
    net.load_state_dict(synth_state_dict(param_shapes(net), 31))
    net.to(DEV)
    eng = net.engine(DEV)
    for B in (1, 3, 17):
        x = randn(600 + B, B, 3, 16, 16).to(DEV)
        a = eng.forward(x, 0.37)
        b = eng.forward(x, torch.full((B,), 0.37, device=DEV))
        # one embedding row broadcast vs B identical rows: the small linear kernels sum in a batch-size-dependent order (fp32 rounding)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-6)


def test_persistent_conv_spin_limit_is_reported_not_hung(ops):
    """conv3x3_ws_kernel bounds every counter poll (conv_ws.inc.h): when the limit expires the launch must end AND say so.  Forced
    here through the diagnostic switches (conv_ablate bit 32: the loaders never publish kernel row 5; a tiny spin limit): the test op
    returns MI355_ERR_TIMEOUT with a message instead of rc 0 with corrupted activations; the same shape with default switches is
    correct right afterwards (the error word is per call)."""
    from mi355._lib import MI355BackendError, debug_config

    x = randn(811, 64, 128, 32, 32)
    sd = synth_state_dict({"weight": (128, 128, 3, 3), "bias": (128,)}, 812)
    want = F.conv2d(x, sd["weight"], sd["bias"], padding=1)
    for gn in (None, True):     # both loader variants: DMA-only (no prologue) and register-staged (GN + SiLU prologue)
        gnp = None
        ref = want
        if gn:
            gsd = synth_state_dict({"g": (128,), "b": (128,)}, 813)
            gnp = (gsd["g"].to(DEV), gsd["b"].to(DEV))
            ref = F.conv2d(F.silu(unet_ref.group_norm32(x, gsd["g"], gsd["b"])), sd["weight"], sd["bias"], padding=1)
        with pytest.raises(MI355BackendError, match="counter wait"):
            ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], gn=gnp, gn_silu=bool(gn), dtype=_lib.MI355_BF16,
                       debug=debug_config(conv_ablate=32, conv_spin_limit=64))
        got = ops.conv2d(x.to(DEV), sd["weight"], sd["bias"], gn=gnp, gn_silu=bool(gn), dtype=_lib.MI355_BF16).cpu()
        torch.testing.assert_close(got, ref, rtol=3e-2, atol=3e-2)
