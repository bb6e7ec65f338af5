"""Stand-ins for the two un-vendored third-party front-ends the reference's cifar10/ and mnist/ scripts
import, so those scripts' sampler call sites work unchanged on the MI355X backend:

  torchcfm.models.unet.unet.UNetModelWrapper   (cifar10/train_cifar10.py:22,92-101; compute_fid.py:16,39-48)
  torchdyn.core.NeuralODE                      (cifar10/utils_cifar.py:4,34-39; compute_fid.py:14,69-70)

Neither package is in /root/reference (versions unpinned; weight URLs point at torchcfm 1.0.4); their
published behaviour is restated from the reference's call sites: constructor keywords, the
`model(t, x, y=None)` call convention with scalar or [B] `t`, and `trajectory(x, t_span)` returning all
len(t_span) states of a fixed-step Euler integration.  `InPaintModelWrapper` / `SuperResModelWrapper` are
the author's unpublished torchcfm edits (mnist/train_mnist.py:34, train_mnist_hy.py:36); their semantics
(channel-concat of the condition / of the bilinearly upsampled low-res image) are inferred from the
keyword arguments at mnist/utils_mnist.py:97 and mnist/utils_mnist_hy.py:82 - "parity unpinned".
"""
from __future__ import annotations

from typing import Optional

import torch

from image_diffusion.unet import UNetModel
from mi355.ops import default_ops


def _default_mult(image_size):
    table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 2, 2),
             28: (1, 2, 2)}
    if image_size not in table:
        raise ValueError(f"unsupported image size: {image_size}")
    return table[image_size]


class UNetModelWrapper(UNetModel):
    """torchcfm UNetModelWrapper: dim=(C,H,W), attention_resolutions as a string of resolutions."""

    def __init__(self, dim, num_channels, num_res_blocks, channel_mult=None, learn_sigma=False, class_cond=False, num_classes=None,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, dropout=0, resblock_updown=False, use_fp16=False, use_new_attention_order=False,
                 in_channels: Optional[int] = None, precision: Optional[str] = None):
        image_size = dim[-1]
        channel_mult = _default_mult(image_size) if channel_mult is None else tuple(channel_mult)
        attention_ds = tuple(image_size // int(res) for res in str(attention_resolutions).split(","))
        # torchcfm: `num_classes = NUM_CLASSES if class_cond else None`-style gating - a label embedding exists only when a class
        # count is given.  Every reference call site passes class_cond=True WITH num_classes=None (mnist/train_mnist.py:262-267,
        # train_mnist2.py:350-355, train_mnist_hy.py:312-318, train_mnist_hy2.py:313-318), i.e. an unconditional network.
        if class_cond and num_classes is not None:
            raise NotImplementedError("class-conditional label_emb (class_cond=True with num_classes set) is not used by the "
                                      "reference's samplers and is not built")
        super().__init__(image_size=image_size, in_channels=dim[0] if in_channels is None else in_channels,
                         model_channels=num_channels, out_channels=(dim[0] if not learn_sigma else dim[0] * 2),
                         num_res_blocks=num_res_blocks, attention_resolutions=attention_ds, dropout=dropout,
                         channel_mult=channel_mult, num_classes=None, use_checkpoint=use_checkpoint, use_fp16=use_fp16,
                         num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                         use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                         use_new_attention_order=use_new_attention_order, precision=precision)

    def _t(self, t, x):
        if isinstance(t, (int, float)) or (isinstance(t, torch.Tensor) and t.dim() == 0 and not t.is_cuda):
            return float(t)          # one host-side time for the batch: the engine broadcasts it (no device tensor, no ATen kernel)
        t = torch.as_tensor(t, device=x.device).float()
        while t.dim() > 1:
            t = t[:, 0]
        if t.dim() == 0:
            t = t.repeat(x.shape[0])
        return t

    @staticmethod
    def _c(t):
        return t if isinstance(t, float) else t.contiguous()

    @torch.no_grad()
    def forward(self, t, x, y=None, *args, **kwargs):
        if not x.is_cuda:
            return super().forward(x, t)     # raises MI355BackendError (no CPU path)
        return self.engine(x.device).forward(x.float().contiguous(), self._c(self._t(t, x)))


class InPaintModelWrapper(UNetModelWrapper):
    """model.forward(x, t, con=con): the -2-sentinel condition image is concatenated on the channel axis."""

    def __init__(self, dim, *a, **kw):
        super().__init__(dim, *a, in_channels=2 * dim[0], **kw)

    @torch.no_grad()
    def forward(self, x, t, con=None, **kwargs):
        eng = self.engine(x.device)
        return eng.forward(x.float().contiguous(), self._c(self._t(t, x)), cond=con.float().contiguous())


class SuperResModelWrapper(UNetModelWrapper):
    """model.forward(x, t, low_res=low_res): low_res is bilinearly upsampled to x's size and concatenated
    (the guided-diffusion SuperResModel convention)."""

    def __init__(self, dim, *a, **kw):
        super().__init__(dim, *a, in_channels=2 * dim[0], **kw)

    def upsample(self, low_res, size):
        """bilinear, align_corners=False (mnist/utils_mnist_hy.py:18-28 is the matching downsample): the HIP resize kernel"""
        return default_ops.resize_bilinear(low_res.float().contiguous(), size)

    @torch.no_grad()
    def forward(self, x, t, low_res=None, **kwargs):
        up = self.upsample(low_res, (x.shape[2], x.shape[3]))
        eng = self.engine(x.device)
        return eng.forward(x.float().contiguous(), self._c(self._t(t, x)), cond=up)


class NeuralODE:
    """torchdyn.core.NeuralODE front-end: solver="euler" (fixed step) or "dopri5" (adaptive, mi355.ode.Dopri5).

    trajectory(x, t_span) -> Tensor[len(t_span), *x.shape].  Euler with a UNetModelWrapper vector field runs the whole
    integration inside libmi355_sampler (mi355_cfm_euler_sample); any other callable f(t, x[, args]) is driven step by step from
    the host with the HIP Euler-update kernel.  dopri5 is one continuous adaptive solve with dense output at the requested times."""

    def __init__(self, vector_field, solver="euler", sensitivity="adjoint", atol=1e-4, rtol=1e-4, **kwargs):
        if solver not in ("euler", "dopri5"):
            raise NotImplementedError(f"solver={solver!r}: only 'euler' and 'dopri5' are built")
        self.vf = vector_field
        self.solver = solver
        self.atol, self.rtol = atol, rtol

    def to(self, *a, **k):
        return self

    def _call(self, t, x):
        # the library's own wrappers take the host scalar as it is; any other vector field gets the 0-dim tensor torchdyn passes
        tt = float(t) if isinstance(self.vf, UNetModelWrapper) else torch.tensor(float(t), device=x.device, dtype=torch.float32)
        try:
            return self.vf(tt, x)
        except TypeError:
            return self.vf(tt, x, None)  # torchdyn passes `args` to 3-argument vector fields (mnist/utils_mnist2.py:120)

    @torch.no_grad()
    def trajectory(self, x, t_span):
        ts = [float(v) for v in torch.as_tensor(t_span).detach().cpu().tolist()]
        x = x.detach().clone().float().contiguous()
        if self.solver == "dopri5":
            # torchdyn's adaptive path (mnist/utils_mnist.py:63-68): states at every requested time; callers index [-1]
            from mi355.ode import Dopri5

            solver = Dopri5(lambda t, y: [self._call(t, y[0])], self.rtol, self.atol)
            states = solver.integrate_times([x], ts)   # one continuous adaptive solve, dense output at every requested time
            return torch.stack([x] + [s[0] for s in states])
        if type(self.vf) is UNetModelWrapper and x.is_cuda:
            _, traj, _ = self.vf.engine(x.device).cfm_euler(x, ts, keep_traj=True)
            return traj
        out = [x.clone()]
        for k in range(len(ts) - 1):
            t = ts[k] if isinstance(self.vf, UNetModelWrapper) else torch.tensor(ts[k], device=x.device, dtype=torch.float32)
            try:
                v = self.vf(t, x)
            except TypeError:
                v = self.vf(t, x, None)  # torchdyn passes `args` to 3-argument vector fields (mnist/utils_mnist2.py:120)
            default_ops.euler_step_(x, v.float().contiguous(), ts[k + 1] - ts[k])
            out.append(x.clone())
        return torch.stack(out)
