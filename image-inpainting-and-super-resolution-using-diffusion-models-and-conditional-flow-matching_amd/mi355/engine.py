"""UNetEngine: one packed-weight U-Net handle of libmi355_sampler.so plus its workspace.

PyTorch is used for device memory (weight blob, workspace, I/O tensors) and the current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import MI355BackendError, check

_PREC = {"fp32": _lib.MI355_F32, "f32": _lib.MI355_F32, "bf16": _lib.MI355_BF16, "bf16x2": _lib.MI355_BF16X2, "fp16": _lib.MI355_F16, "f16": _lib.MI355_F16}


def param_inventory(cfg: _lib.UNetConfigC):
    """(name, shape) list from the C++ plan builder, reference state_dict order."""
    L = _lib.lib()
    n = check(L.mi355_unet_param_count(C.byref(cfg)), "mi355_unet_param_count")
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    nd = C.c_int()
    for i in range(n):
        check(L.mi355_unet_param_info(C.byref(cfg), i, name, 256, shape, C.byref(nd)))
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value))))
    return out


class UNetEngine:
    def __init__(self, cfg_kwargs: dict, state_dict: Dict[str, torch.Tensor], device, precision: str = "bf16", differentiable: bool = False,
                 debug=None):
        if precision not in _PREC:
            raise ValueError(f"precision must be one of {sorted(_PREC)}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise MI355BackendError(f"UNetEngine needs an MI355X device, got {self.device} (no CPU fallback)")
        self.precision = precision
        self.differentiable = bool(differentiable)
        self.cfg = _lib.make_config(dtype=_PREC[precision], differentiable=differentiable, debug=debug, **cfg_kwargs)
        self.L = _lib.lib()
        inv = param_inventory(self.cfg)
        host = []
        for name, shape in inv:
            if name not in state_dict:
                raise KeyError(f"state_dict is missing {name}")
            t = state_dict[name].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
            host.append(t)
        wbytes = check(self.L.mi355_unet_weight_bytes(C.byref(self.cfg)), "mi355_unet_weight_bytes")
        with torch.cuda.device(self.device):
            self.weights = torch.empty(wbytes, dtype=torch.uint8, device=self.device)
            ptrs = (C.c_void_p * len(host))(*[t.data_ptr() for t in host])
            handle = C.c_void_p()
            check(self.L.mi355_unet_create(C.byref(self.cfg), ptrs, len(host), C.c_void_p(self.weights.data_ptr()), wbytes,
                                           self._stream(), C.byref(handle)), "mi355_unet_create")
        self.handle = handle
        self._ws: Optional[torch.Tensor] = None
        self._fwd_state = None   # (batch, workspace pointer) of the last forward(): what vjp() differentiates
        self.in_channels = self.cfg.in_channels
        self.out_channels = self.cfg.out_channels
        self.image_size = self.cfg.image_size

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.L.mi355_unet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def check(self, clear: bool = True):
        """Raise MI355BackendError if a launch of this engine gave up a bounded counter wait (mi355_unet_status).  Synchronise first to
        cover the launches already queued; every engine call also checks the flag on entry."""
        check(self.L.mi355_unet_status(self.handle, int(clear)), "mi355_unet_status")

    def workspace(self, batch: int):
        need = check(self.L.mi355_unet_workspace_bytes(self.handle, batch), "mi355_unet_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return C.c_void_p(self._ws.data_ptr()), self._ws.numel()

    def _chk(self, t: torch.Tensor, name: str, dtype=torch.float32):
        if not t.is_cuda or t.device != self.device:
            raise MI355BackendError(f"{name} is on {t.device}, the engine lives on {self.device} (no CPU fallback)")
        if t.dtype != dtype or not t.is_contiguous():
            raise TypeError(f"{name} must be contiguous {dtype}")
        return C.c_void_p(t.data_ptr())

    def _split(self, x, cond):
        B, Cx, H, W = x.shape
        if H != self.image_size or W != self.image_size:
            raise ValueError(f"expected {self.image_size}x{self.image_size} images, got {H}x{W}")
        Cc = 0
        if cond is not None:
            if cond.shape[0] != B or cond.shape[2:] != x.shape[2:]:
                raise ValueError("condition must match x in batch and spatial size")
            Cc = cond.shape[1]
        if Cx + Cc != self.in_channels:
            raise ValueError(f"x ({Cx}) + condition ({Cc}) channels != in_channels ({self.in_channels})")
        return B, Cx, Cc

    def forward(self, x: torch.Tensor, t, cond: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """t: a [B] device tensor, or a host scalar shared by the batch (no device tensor is made for it: mi355_unet_forward_t)."""
        B, Cx, Cc = self._split(x, cond)
        if isinstance(t, (int, float)):
            if out is None:
                out = torch.empty(B, self.out_channels, self.image_size, self.image_size, device=self.device, dtype=torch.float32)
            ws, wsb = self.workspace(B)
            check(self.L.mi355_unet_forward_t(self.handle, self._chk(x, "x"), Cx, self._chk(cond, "condition") if cond is not None else None,
                                              Cc, float(t), self._chk(out, "out"), B, ws, wsb, self._stream()), "mi355_unet_forward_t")
            self._fwd_state = (B, self._ws.data_ptr())
            return out
        if t.shape != (B,):
            raise ValueError(f"timesteps must have shape ({B},)")
        if out is None:
            out = torch.empty(B, self.out_channels, self.image_size, self.image_size, device=self.device, dtype=torch.float32)
        ws, wsb = self.workspace(B)
        check(self.L.mi355_unet_forward(self.handle, self._chk(x, "x"), Cx, self._chk(cond, "condition") if cond is not None else None,
                                        Cc, self._chk(t, "timesteps"), self._chk(out, "out"), B, ws, wsb, self._stream()),
              "mi355_unet_forward")
        self._fwd_state = (B, self._ws.data_ptr())
        return out

    def vjp(self, grad_out: torch.Tensor, x_channels: Optional[int] = None, out: Optional[torch.Tensor] = None):
        """(d out / d x)^T grad_out of the LAST forward() on this engine (same batch): the reconstruction-guidance gradient
        (AD/image_diffusion/sampling.py:154-163).  Needs an engine built with differentiable=True."""
        if not self.differentiable:
            raise MI355BackendError("vjp needs an engine built with differentiable=True")
        B = grad_out.shape[0]
        Cx = self.out_channels if x_channels is None else int(x_channels)
        if tuple(grad_out.shape) != (B, self.out_channels, self.image_size, self.image_size):
            raise ValueError("grad_out must have the shape of the network output")
        if out is None:
            out = torch.empty(B, Cx, self.image_size, self.image_size, device=self.device, dtype=torch.float32)
        ws, wsb = self.workspace(B)
        if self._fwd_state != (B, self._ws.data_ptr()):
            # the arena offsets scale with the batch and the samplers / profile() reuse the workspace: anything but a forward() of
            # this batch as the last call leaves other activations there, and the gradient would be garbage with rc 0
            raise MI355BackendError("vjp: the last call on this engine was not forward() with the same batch (no activations to differentiate)")
        check(self.L.mi355_unet_vjp(self.handle, self._chk(grad_out, "grad_out"), self._chk(out, "grad_x"), Cx, B, ws, wsb, self._stream()),
              "mi355_unet_vjp")
        return out

    def plan_ops(self):
        """Diagnostics: the plan as a list of dicts (op kinds: 0 GN, 1 conv, 2 attention, 3 resample, 4 pool-affine, 5 fused attention)."""
        names = ("kind", "src0", "src1", "dst", "mode", "ks", "cout", "use_pro", "pro_silu", "res", "res_mode", "gn_site", "heads", "ch", "dst_c", "dst_h")
        buf = (C.c_int32 * 16)()
        out, i = [], 0
        while True:
            n = self.L.mi355_unet_plan_op(self.handle, i, buf)
            if n < 0:
                break
            out.append(dict(zip(names, list(buf))))
            i += 1
            if i >= n:
                break
        return out

    def read_tensor(self, tensor: int, batch: int, shape, gradient: bool = False):
        """Diagnostics: activation (or gradient) `tensor` of the last forward (vjp) as NCHW fp32."""
        out = torch.empty((batch,) + tuple(shape), device=self.device, dtype=torch.float32)
        ws, wsb = self.workspace(batch)
        check(self.L.mi355_unet_read_tensor(self.handle, int(tensor), int(gradient), self._chk(out, "out"), batch, ws, wsb, self._stream()),
              "mi355_unet_read_tensor")
        return out

    def profile(self, x: torch.Tensor, t: torch.Tensor, cond: Optional[torch.Tensor] = None):
        """One forward with HIP events around every op -> list of dicts (kind, ks, cin, cout, h, w, tile, ms, flops, bytes)."""
        B, Cx, Cc = self._split(x, cond)
        out = torch.empty(B, self.out_channels, self.image_size, self.image_size, device=self.device, dtype=torch.float32)
        self._fwd_state = None
        ws, wsb = self.workspace(B)
        cap = 4096
        recs = (_lib.OpProfileC * cap)()
        n = check(self.L.mi355_unet_profile(self.handle, self._chk(x, "x"), Cx, self._chk(cond, "condition") if cond is not None else None,
                                            Cc, self._chk(t, "timesteps"), self._chk(out, "out"), B, ws, wsb, self._stream(), recs, cap),
                  "mi355_unet_profile")
        names = {0: "prelude", 1: "gn_stats", 2: "conv", 3: "attention", 4: "resample"}
        return [dict(kind=names[r.kind], ks=r.ks, cin=r.cin, cout=r.cout, h=r.h, w=r.w, tile=(r.tile_m, r.tile_n), ms=r.ms,
                     flops=r.flops, bytes=r.bytes) for r in recs[:min(n, cap)]]

    def stats(self, batch: int):
        s = _lib.UNetStatsC()
        check(self.L.mi355_unet_get_stats(self.handle, batch, C.byref(s)))
        return {"launches": s.launches, "conv_flops": s.conv_flops, "attn_flops": s.attn_flops, "act_bytes": s.act_bytes,
                "weight_bytes": s.weight_bytes}

    def max_batch(self) -> int:
        """Largest batch one call can take: the kernels address every activation tensor through 32-bit buffer offsets, so B x (the largest
        tensor of the plan, per image) must stay under 4 GiB (the library refuses larger launches with "run the batch in slices")."""
        if getattr(self, "max_batch_override", None):
            return int(self.max_batch_override)
        if getattr(self, "_max_batch", None) is None:
            esz = 4 if self.precision in ("fp32", "f32") else 2
            per_image = max([op["dst_c"] * op["dst_h"] * op["dst_h"] * esz for op in self.plan_ops() if op["dst"] >= 0] +
                            [32 * self.image_size * self.image_size * 4])   # the samplers' fp32 scratch holds up to 32 channels
            self._max_batch = max(1, 0xFFFF0000 // (2 * per_image))   # x2: a concat source pair / an in-flight double of the same tensor
        return self._max_batch

    def cfm_euler(self, x: torch.Tensor, t_span: Sequence[float], cond: Optional[torch.Tensor] = None, keep_traj: bool = False,
                  want_u8: bool = False, cond_drift: bool = False):
        """In-place Euler integration of x over t_span (host floats).  Returns (x, traj or None, u8 or None).
        cond_drift: the condition is integrated with derivative `cond` (the concatenated-state sampler of
        mnist/utils_mnist2.py:118-138); the caller's tensor is not modified.
        A batch beyond max_batch() is integrated in slices (every image's trajectory is independent of its batch mates; the kernels chosen for a
        slice may sum in another order than those of the whole batch would)."""
        B, Cx, Cc = self._split(x, cond)
        mb = self.max_batch()
        if B > mb:
            traj = torch.empty((len(t_span),) + tuple(x.shape), device=self.device, dtype=torch.float32) if keep_traj else None
            u8 = torch.empty(x.shape, device=self.device, dtype=torch.uint8) if want_u8 else None
            for lo in range(0, B, mb):
                hi = min(B, lo + mb)
                _, tr, u = self.cfm_euler(x[lo:hi], t_span, cond[lo:hi] if cond is not None else None, keep_traj, want_u8, cond_drift)
                if traj is not None:
                    traj[:, lo:hi] = tr
                if u8 is not None:
                    u8[lo:hi] = u
            return x, traj, u8
        ts = [float(v) for v in t_span]
        arr = (C.c_float * len(ts))(*ts)
        traj = torch.empty((len(ts),) + tuple(x.shape), device=self.device, dtype=torch.float32) if keep_traj else None
        u8 = torch.empty(x.shape, device=self.device, dtype=torch.uint8) if want_u8 else None
        self._fwd_state = None
        ws, wsb = self.workspace(B)
        check(self.L.mi355_cfm_euler_sample(self.handle, self._chk(x, "x"), Cx, self._chk(cond, "condition") if cond is not None else None,
                                            Cc, int(bool(cond_drift)), arr, len(ts), self._chk(traj, "traj") if traj is not None else None,
                                            self._chk(u8, "u8", torch.uint8) if u8 is not None else None, B, ws, wsb, self._stream()),
              "mi355_cfm_euler_sample")
        return x, traj, u8

    def ddpm_sample(self, x: torch.Tensor, tables: Dict[str, torch.Tensor], *, mode: int, cond: Optional[torch.Tensor] = None,
                    noise: Optional[torch.Tensor] = None, n_corrector=0, delta=0.1, tmin=1e-5, tmax=1.0, start_fraction=1.0,
                    noise_condition=True, pad_value=-2.0, none_value=-2.0, seed=0):
        """In-place reverse-denoising loop.  tables: name -> CPU fp32 tensor [Ns] (DDPM buffers)."""
        B, Cx = x.shape[:2]
        if cond is not None and cond.shape != x.shape:
            raise ValueError("condition must have the shape of x")
        tb = _lib.DDPMTablesC()
        keep = []
        Ns = None
        fp = C.POINTER(C.c_float)
        for name, _ in _lib.DDPMTablesC._fields_[1:]:
            v = tables[name].detach().to("cpu", torch.float32).contiguous()
            Ns = v.numel() if Ns is None else Ns
            if v.numel() != Ns:
                raise ValueError("DDPM tables must all have length Ns")
            keep.append(v)
            setattr(tb, name, C.cast(v.data_ptr(), fp))
        tb.Ns = Ns
        opt = _lib.DDPMOptionsC(mode, n_corrector, delta, tmin, tmax, start_fraction, int(noise_condition), pad_value, none_value,
                                int(cond is None), seed)
        ndraws = 0
        if noise is not None:
            if noise.shape[1:] != x.shape:
                raise ValueError("injected noise must be [n_draws, B, C, H, W]")
            ndraws = noise.shape[0]
        self._fwd_state = None
        ws, wsb = self.workspace(B)
        check(self.L.mi355_ddpm_sample(self.handle, self._chk(x, "x"), Cx, self._chk(cond, "condition") if cond is not None else None,
                                       C.byref(tb), C.byref(opt), self._chk(noise, "noise") if noise is not None else None, ndraws, B,
                                       ws, wsb, self._stream()), "mi355_ddpm_sample")
        return x
