"""Tensor-level wrappers of the single-op C entry points (PyTorch supplies memory and the stream only).

Every function validates its operands on the host (device, dtype, contiguity, shapes) before a
hand-written kernel is launched, enqueues on torch's current HIP stream and never synchronises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import MI355BackendError, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise MI355BackendError(f"{name} is on {t.device}: the MI355X HIP backend needs device tensors (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


def _same(a, b, na, nb):
    if a.shape != b.shape:
        raise ValueError(f"{na} {tuple(a.shape)} and {nb} {tuple(b.shape)} must have the same shape")


class Ops:
    """Default op table used by the samplers (tests may inject a recording double with the same methods)."""

    def timestep_embedding(self, t, dim, max_period=10000.0):
        out = torch.empty(t.shape[0], dim, device=t.device, dtype=torch.float32)
        check(_lib.lib().mi355_timestep_embedding(_req(t, "t"), t.shape[0], dim, float(max_period), _req(out, "out"), _stream()))
        return out

    def groupnorm(self, x, gamma, beta, groups=32, eps=1e-5, silu=False):
        B, Cc = x.shape[:2]
        hw = x[0, 0].numel()
        y = torch.empty_like(x)
        check(_lib.lib().mi355_groupnorm(_req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta"), _req(y, "y"), B, Cc, hw, groups,
                                         float(eps), int(silu), _stream()))
        return y

    def euler_step_(self, x, v, dt):
        _same(x, v, "x", "v")
        check(_lib.lib().mi355_euler_step(_req(x, "x"), _req(v, "v"), float(dt), x.numel(), _stream()))
        return x

    def ddpm_step_(self, x, eps, z, c_recip, c_recipm1, coef1, coef2, sigma, philox=None):
        """z: injected noise tensor, or None; philox: (seed, offset) for device noise; both None = no noise (i == 0)."""
        _same(x, eps, "x", "eps")
        zp = _req(z, "z") if z is not None else None
        if z is not None:
            _same(x, z, "x", "z")
        seed, off = philox if philox else (0, 0)
        check(_lib.lib().mi355_ddpm_step(_req(x, "x"), _req(eps, "eps"), zp, c_recip, c_recipm1, coef1, coef2, sigma,
                                         int(philox is not None and z is None), seed, off, x.numel(), _stream()))
        return x

    def corrector_step_(self, x, eps, z, c_recip, c_recipm1, rsm1, dt, delta, philox=None):
        _same(x, eps, "x", "eps")
        zp = _req(z, "z") if z is not None else None
        seed, off = philox if philox else (0, 0)
        check(_lib.lib().mi355_corrector_step(_req(x, "x"), _req(eps, "eps"), zp, c_recip, c_recipm1, rsm1, dt, delta,
                                              int(philox is not None and z is None), seed, off, x.numel(), _stream()))
        return x

    def ddim_step_(self, x, eps, c_recip, c_recipm1, acp_prev):
        _same(x, eps, "x", "eps")
        check(_lib.lib().mi355_ddim_step(_req(x, "x"), _req(eps, "eps"), c_recip, c_recipm1, acp_prev, x.numel(), _stream()))
        return x

    def replace_mask_(self, x, cond, z, pad_value, noisy, sa, sb, philox=None):
        _same(x, cond, "x", "condition")
        zp = _req(z, "z") if z is not None else None
        seed, off = philox if philox else (0, 0)
        check(_lib.lib().mi355_replace_mask(_req(x, "x"), _req(cond, "condition"), zp, float(pad_value), int(noisy), sa, sb,
                                            int(philox is not None and z is None), seed, off, x.numel(), _stream()))
        return x

    def guidance_seed(self, x, eps, cond, c_recip, c_recipm1, mode, pad_value):
        """-> (g_eps, g_x): cotangent for the U-Net VJP and the direct-path gradient of the per-sample constraint
        (mode 0 = Painting.loss with the pad sentinel masked, 1 = HyperResolution.loss; sampling.py:148-160)."""
        _same(x, eps, "x", "eps")
        _same(x, cond, "x", "condition")
        g_eps, g_x = torch.empty_like(x), torch.empty_like(x)
        check(_lib.lib().mi355_guidance_seed(_req(x, "x"), _req(eps, "eps"), _req(cond, "condition"), float(c_recip), float(c_recipm1),
                                             int(mode), float(pad_value), x[0].numel(), _req(g_eps, "g_eps"), _req(g_x, "g_x"), x.numel(),
                                             _stream()))
        return g_eps, g_x

    def guidance_update_(self, x, g_x, vjp, scale, apply):
        """update = -scale * (g_x + vjp); x += update when `apply` (the "before" rule).  -> update"""
        _same(x, g_x, "x", "g_x")
        _same(x, vjp, "x", "vjp")
        upd = torch.empty_like(x)
        check(_lib.lib().mi355_guidance_update(_req(x, "x"), _req(g_x, "g_x"), _req(vjp, "vjp"), float(scale), int(bool(apply)),
                                               _req(upd, "update"), x.numel(), _stream()))
        return upd

    def clip_(self, x, lo=-1.0, hi=1.0):
        check(_lib.lib().mi355_clip(_req(x, "x"), float(lo), float(hi), x.numel(), _stream()))
        return x

    def ema_update_(self, target, source, decay):
        """target = target * decay + source * (1 - decay), in place (cifar10/utils_cifar.py:47-53)."""
        _same(target, source, "target", "source")
        check(_lib.lib().mi355_ema_update(_req(target, "target"), _req(source, "source"), float(decay), float(1 - decay), target.numel(), _stream()))
        return target

    def mse_per_sample(self, a, b):
        """torch.mean((a - b)**2, dim=(1, 2, 3))  (AD/experiments/main.py:299)."""
        _same(a, b, "a", "b")
        out = torch.empty(a.shape[0], device=a.device, dtype=torch.float32)
        check(_lib.lib().mi355_mse_per_sample(_req(a, "a"), _req(b, "b"), _req(out, "out"), a.shape[0], a[0].numel(), _stream()))
        return out

    def lincomb_per_sample(self, x, a, y=None, b=None, out=None):
        """out[n] = a[n] * x[n] (+ b[n] * y[n]) with per-sample fp32 device coefficients a, b of shape [B] (sde_diffusion.py:214-244)."""
        B = x.shape[0]
        if a.shape != (B,) or (b is not None and b.shape != (B,)):
            raise ValueError("per-sample coefficients must have shape [B]")
        if (y is None) != (b is None):
            raise ValueError("y and b go together")
        if y is not None:
            _same(x, y, "x", "y")
        out = torch.empty_like(x) if out is None else out
        check(_lib.lib().mi355_lincomb_per_sample(_req(out, "out"), _req(x, "x"), _req(y, "y") if y is not None else None, _req(a, "a"),
                                                  _req(b, "b") if b is not None else None, B, x[0].numel(), _stream()))
        return out

    def resize_bilinear(self, x, size):
        """F.interpolate(x, size=size, mode="bilinear", align_corners=False) on an NCHW fp32 device tensor (HIP kernel)."""
        if x.dim() != 4:
            raise ValueError("resize_bilinear expects an NCHW tensor")
        Ho, Wo = int(size[0]), int(size[1])
        out = torch.empty(x.shape[0], x.shape[1], Ho, Wo, device=x.device, dtype=torch.float32)
        check(_lib.lib().mi355_resize_bilinear(_req(x, "x"), _req(out, "out"), x.shape[0] * x.shape[1], x.shape[2], x.shape[3], Ho, Wo,
                                               _stream()), "mi355_resize_bilinear")
        return out

    def paint_patch(self, images, top, left, patch_size, pad_value, outpaint=False):
        """InPainting / OutPainting condition of a whole batch: window (top[n], left[n]) of image n (int32 device tensors [N])."""
        N, Cc, H, W = images.shape
        if tuple(top.shape) != (N,) or tuple(left.shape) != (N,):
            raise ValueError("top / left must have one entry per image")
        out = torch.empty_like(images)
        check(_lib.lib().mi355_paint_patch(_req(images, "images"), _req(top, "top", torch.int32), _req(left, "left", torch.int32),
                                           int(patch_size), float(pad_value), int(bool(outpaint)), _req(out, "out"), N, Cc, H, W, _stream()),
              "mi355_paint_patch")
        return out

    def quantize_u8(self, x):
        out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        check(_lib.lib().mi355_quantize_u8(_req(x, "x"), _req(out, "out", torch.uint8), x.numel(), _stream()))
        return out

    def to_unit_range(self, x):
        out = torch.empty_like(x)
        check(_lib.lib().mi355_to_unit_range(_req(x, "x"), _req(out, "out"), x.numel(), _stream()))
        return out

    def randn(self, shape, device, seed, offset=0):
        out = torch.empty(shape, device=device, dtype=torch.float32)
        check(_lib.lib().mi355_randn(_req(out, "out"), int(seed), int(offset), out.numel(), _stream()))
        return out

    # --- adaptive RK45 building blocks (csrc/ode.hip) ---
    def rk_combine(self, out, y0, ks, coeffs):
        """out = y0 + sum_j coeffs[j] * ks[j]  (y0 may be None; coeffs already include dt)."""
        assert len(ks) == len(coeffs) <= 7
        arr = (C.c_float * 7)(*([float(c) for c in coeffs] + [0.0] * (7 - len(coeffs))))
        kp = [_req(k, "k") for k in ks] + [None] * (7 - len(ks))
        check(_lib.lib().mi355_rk_combine(_req(out, "out"), _req(y0, "y0") if y0 is not None else None, *kp, arr, len(ks), out.numel(),
                                          _stream()))
        return out

    def rk_sqnorm(self, acc, a, sub=None, b=None, b2=None, atol=1.0, rtol=0.0):
        """acc (device fp64 scalar tensor) += sum(((a - sub) / (atol + rtol * max(|b|, |b2|)))**2)."""
        check(_lib.lib().mi355_rk_sqnorm(_req(a, "a"), _req(sub, "sub") if sub is not None else None, _req(b, "b") if b is not None else None,
                                         _req(b2, "b2") if b2 is not None else None, float(atol), float(rtol), a.numel(),
                                         _req(acc, "acc", torch.float64), _stream()))
        return acc

    def rk_interp(self, out, y0, y1, ymid, f0, f1, dt, x):
        check(_lib.lib().mi355_rk_interp(_req(out, "out"), _req(y0, "y0"), _req(y1, "y1"), _req(ymid, "ymid"), _req(f0, "f0"), _req(f1, "f1"),
                                         float(dt), float(x), out.numel(), _stream()))
        return out

    # --- parity-test ops on NCHW fp32 tensors (pack -> MFMA kernel -> unpack) ---
    def conv2d(self, x, weight, bias=None, stride=1, resample=0, gn=None, gn_silu=False, dtype=_lib.MI355_F32, x1=None, emb=None,
               res=None, res_mode=1, debug=None):
        """weight/bias: CPU fp32 tensors in the reference layout [Co,Ci(+Ci1),k,k]; gn = (gamma, beta) device tensors over the
        (concatenated) input channels; x1: second source of a channel concat; emb [B, Co]; res [B, Co, Hr, Wr] with res_mode 1
        (same size) or 2 (nearest x2 of a half-size tensor)."""
        B, Cin, H, W = x.shape
        Co, Ci, k, _ = weight.shape
        Cin1 = 0 if x1 is None else x1.shape[1]
        assert Ci == Cin + Cin1
        if x1 is not None and (x1.shape[0] != B or x1.shape[2:] != x.shape[2:]):
            raise ValueError("x1 must match x in batch and spatial size")
        Hc = H * 2 if resample == 2 else (H // 2 if resample == 3 else H)
        Wc = W * 2 if resample == 2 else (W // 2 if resample == 3 else W)
        Ho = (Hc + 2 * (k // 2) - k) // stride + 1
        Wo = (Wc + 2 * (k // 2) - k) // stride + 1
        if emb is not None and emb.shape != (B, Co):
            raise ValueError("emb must be [B, Co]")
        if res is not None:
            want = (B, Co, Ho, Wo) if res_mode == 1 else (B, Co, Ho // 2, Wo // 2)
            if tuple(res.shape) != want:
                raise ValueError(f"res must be {want}")
        y = torch.empty(B, Co, Ho, Wo, device=x.device, dtype=torch.float32)
        L = _lib.lib()
        wsb = L.mi355_op_workspace_bytes(B, max(Ci, Co), max(H * W, Ho * Wo))
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous() if bias is not None else None
        fp = C.POINTER(C.c_float)
        check(L.mi355_conv2d(_req(x, "x"), _req(x1, "x1") if x1 is not None else None, Cin1, C.cast(w.data_ptr(), fp),
                             C.cast(b.data_ptr(), fp) if b is not None else None, _req(y, "y"), B, Cin, H, W, Co, k, stride, resample,
                             _req(gn[0], "gamma") if gn else None, _req(gn[1], "beta") if gn else None, int(gn_silu),
                             _req(emb, "emb") if emb is not None else None, _req(res, "res") if res is not None else None, int(res_mode),
                             dtype, C.byref(debug if debug is not None else _lib.debug_config()), C.c_void_p(ws.data_ptr()), wsb, _stream()),
              "mi355_conv2d")
        return y

    def conv2d_ex(self, x, weight, bias, dtype=_lib.MI355_F32, skip=None, sites=(), film=None, debug=None):
        """The 3x3 conv of the 8x8 / 4x4 levels with its fused forms (mi355_conv2d_ex): skip = (xs0, xs1 or None, w1 [Co, c0 + c1, 1, 1], b1) is a
        1x1 conv of cat(xs0, xs1) accumulated into the same output; sites = up to two dicts (ctotal, coff, gamma, beta, silu): GroupNorm32 (+SiLU)
        of the output as channels coff.. of a ctotal-channel tensor, written by the epilogue; film [B, 2 Co] on site 0.  Returns
        (y, [act or None per site], skip_done)."""
        B, Cin, H, W = x.shape
        Co, Ci, k, _ = weight.shape
        assert Ci == Cin and k == 3 and len(sites) <= 2
        y = torch.empty(B, Co, H, W, device=x.device, dtype=torch.float32)
        L = _lib.lib()
        wsb = L.mi355_op_workspace_bytes(B, max(Ci, Co), H * W)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous() if bias is not None else None
        fp = C.POINTER(C.c_float)
        ex = _lib.ConvExtrasC()
        keep = []
        if skip is not None:
            xs0, xs1, w1, b1 = skip
            w1c = w1.detach().to("cpu", torch.float32).reshape(Co, -1).contiguous()
            b1c = b1.detach().to("cpu", torch.float32).contiguous() if b1 is not None else None
            keep += [w1c, b1c]
            ex.skip_x0 = _req(xs0, "skip_x0"); ex.skip_c0 = xs0.shape[1]
            if xs1 is not None:
                ex.skip_x1 = _req(xs1, "skip_x1"); ex.skip_c1 = xs1.shape[1]
            ex.skip_w_host = w1c.data_ptr()
            ex.skip_bias_host = b1c.data_ptr() if b1c is not None else None
        acts = []
        for i, st in enumerate(sites):
            a = torch.full((B, st["ctotal"], H, W), float("nan"), device=x.device, dtype=torch.float32)
            acts.append(a)
            ex.act_out[i] = _req(a, "act_out"); ex.act_gamma[i] = _req(st["gamma"], "gamma"); ex.act_beta[i] = _req(st["beta"], "beta")
            ex.act_ctotal[i] = st["ctotal"]; ex.act_coff[i] = st["coff"]; ex.act_silu[i] = int(st.get("silu", True))
        if film is not None:
            ex.act_film = _req(film, "film")
        check(L.mi355_conv2d_ex(_req(x, "x"), None, 0, C.cast(w.data_ptr(), fp), C.cast(b.data_ptr(), fp) if b is not None else None, _req(y, "y"),
                                B, Cin, H, W, Co, 3, 1, 0, None, None, 0, None, None, 1, dtype,
                                C.byref(debug if debug is not None else _lib.debug_config()), C.c_void_p(ws.data_ptr()), wsb, _stream(), C.byref(ex)),
              "mi355_conv2d_ex")
        return y, [a if (ex.act_done >> i) & 1 else None for i, a in enumerate(acts)], bool(ex.skip_done)

    def qkv_attention(self, qkv, heads, new_order=False, dtype=_lib.MI355_F32):
        B, width, T = qkv.shape
        ch = width // (3 * heads)
        out = torch.empty(B, heads * ch, T, device=qkv.device, dtype=torch.float32)
        L = _lib.lib()
        wsb = L.mi355_op_workspace_bytes(B, width, T)
        ws = torch.empty(wsb, device=qkv.device, dtype=torch.uint8)
        check(L.mi355_qkv_attention(_req(qkv, "qkv"), _req(out, "out"), B, heads, ch, T, int(new_order), dtype,
                                    C.c_void_p(ws.data_ptr()), wsb, _stream()), "mi355_qkv_attention")
        return out


default_ops = Ops()
