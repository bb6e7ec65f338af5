#!/usr/bin/env python3
"""Experiment (round 5, last): the bench batch as TWO independent half-batch chains on two HIP streams (each chain's persistent kernels then ask for
half the CUs: do the chains fill each other's launch ramps and drains?) against the one-chain default.  Same weights, same x0, interleaved repeats.
  python tools/experiments/scripts/r5_two_chains.py [B] [reps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R)
import tests.conftest  # noqa: F401  (package path)
import torch
from image_diffusion.unet import UNetModel, param_shapes
from mi355.synth import synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
kw = dict(image_size=32, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(2,), channel_mult=(1, 2, 2, 2),
          num_heads=4, num_head_channels=64)
sd = None


def engine():
    global sd
    net = UNetModel(precision="bf16", **kw)
    if sd is None:
        sd = synth_state_dict(param_shapes(net), 1234)
    net.load_state_dict(sd)
    net.to(dev)
    return net, net.engine(dev)


n0, e0 = engine()
n1, e1 = engine()
n2, e2 = engine()
x0 = torch.randn(B, 3, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
ts = torch.linspace(0, 1, 51).tolist()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    x = x0.clone()
    e0.cfm_euler(x, ts, want_u8=True)
    return x


def two(parts=2):
    x = x0.clone()
    torch.cuda.current_stream().synchronize()
    h = B // 2
    with torch.cuda.stream(sa):
        e1.cfm_euler(x[:h], ts, want_u8=True)
    with torch.cuda.stream(sb):
        e2.cfm_euler(x[h:], ts, want_u8=True)
    sa.synchronize(); sb.synchronize()
    return x


ya, yb = one(), two()
torch.cuda.synchronize()
print("max |one - two| =", (ya - yb).abs().max().item())
for r in range(reps):
    for name, f in (("one chain ", one), ("two chains", two)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{name}: {B / dt:8.1f} images/s  {dt * 1e3:7.2f} ms/step", flush=True)
