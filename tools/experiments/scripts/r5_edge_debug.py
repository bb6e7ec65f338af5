# which of the two edge fusions (conv_edge bits 2 / 3) changes bits, and is the loop deterministic at all?
import sys, torch
sys.path.insert(0, "/root/repo/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd")
from image_diffusion.unet import UNetModel, param_shapes
from mi355._lib import debug_config
from mi355.synth import randn, synth_state_dict
DEV = "cuda:0"
kw = dict(image_size=32, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=1, attention_resolutions=(2,), channel_mult=(1, 2), num_heads=4, num_head_channels=64)
sd = None
def run(prec, steps, **knobs):
    global sd
    net = UNetModel(precision=prec, **kw)
    if sd is None: sd = synth_state_dict(param_shapes(net), 5501)
    net.load_state_dict(sd); net.debug = debug_config(**knobs); net.to(DEV)
    x = randn(5500, 18, 3, 32, 32).to(DEV)
    e = net.engine(DEV)
    y, _, _ = e.cfm_euler(x.clone(), steps)
    f = e.forward(x, torch.linspace(0, 1, 18).to(DEV))
    torch.cuda.synchronize(); e.check()
    return y.cpu(), f.cpu()
for prec in ("bf16", "fp32"):
    for steps in ([0.0, 1.0], [0.0, 0.2, 0.5, 0.6, 1.0]):
        base = run(prec, steps, conv_edge=3)
        for ce in (3, 7, 11, 15):
            r = run(prec, steps, conv_edge=ce)
            print(prec, len(steps) - 1, "steps conv_edge", ce, "euler maxdiff", (r[0] - base[0]).abs().max().item(), "forward maxdiff", (r[1] - base[1]).abs().max().item(), flush=True)
