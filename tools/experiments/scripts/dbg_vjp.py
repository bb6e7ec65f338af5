#!/usr/bin/env python3
"""Debug: per-tensor gradients of the differentiable plan vs torch.autograd on the oracle's functional U-Net (tiny net)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd")
sys.path[:0] = [REPO, PKG]
import torch
from image_diffusion.unet import UNetModel, param_shapes
from mi355.synth import randn, synth_state_dict
from oracle import unet_ref

dev = "cuda:0"
cfg = unet_ref.UNetConfig(16, 1, 32, 1, 1, (2,), channel_mult=(1, 2), num_heads=2)
net = UNetModel(image_size=16, in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2),
                num_heads=2, precision="fp32")
sd = synth_state_dict(param_shapes(cfg), 1001)
net.load_state_dict(sd); net.to(dev)
eng = net.engine(dev, differentiable=True)
B = 2
x, t, g = randn(1, B, 1, 16, 16), torch.tensor([0.37, 0.91]), randn(2, B, 1, 16, 16)
y = eng.forward(x.to(dev), t.to(dev))
gx = eng.vjp(g.to(dev), x_channels=1)
xr = x.clone().requires_grad_()
yr = unet_ref.unet_forward_diff(sd, cfg, xr, t)
(ref,) = torch.autograd.grad((yr * g).sum(), xr)
print("forward err", (y.cpu() - yr.detach()).abs().max().item(), "grad err", (gx.cpu() - ref).abs().max().item(), "ref max", ref.abs().max().item())
ops = eng.plan_ops()
for i, o in enumerate(ops):
    if o["dst"] >= 0:
        gt = eng.read_tensor(o["dst"], B, (o["dst_c"], o["dst_h"], o["dst_h"]), gradient=True).cpu()
        at = eng.read_tensor(o["dst"], B, (o["dst_c"], o["dst_h"], o["dst_h"])).cpu()
        print(i, o["kind"], "src", o["src0"], o["src1"], "dst", o["dst"], "C", o["dst_c"], "H", o["dst_h"], "pro", o["use_pro"], "res", o["res"],
              "| act absmax %.3f nz %.2f | grad absmax %.3e nz-frac %.3f" % (at.abs().max(), (at != 0).float().mean(), gt.abs().max(), (gt != 0).float().mean()))
    else:
        print(i, o["kind"], "src", o["src0"], o["src1"], "dst", o["dst"], "pro", o["use_pro"])
g0 = eng.read_tensor(0, B, (16, 16, 16), gradient=True).cpu()
print("grad of input tensor: nz-frac per channel", [(g0[:, c] != 0).float().mean().item() for c in range(4)])
