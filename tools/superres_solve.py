#!/usr/bin/env python3
"""One `utils_mnist_hy2.generate_samples_eval` super-resolution solve (mnist/utils_mnist_hy2.py:76-98: SuperResModelWrapper, the low-res
condition drifts with the state) with synthetic weights - the command whose rocprofv3 kernel summary shows that no `at::native`
kernel runs per network evaluation (VERDICT r2 task 8):

    rocprofv3 --kernel-trace --stats -d gpurun_out/sr -o s -- python3 tools/superres_solve.py --steps 40
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--solver", default="euler")
    a = ap.parse_args()
    import utils_mnist_hy2
    from image_diffusion.unet import param_shapes
    from mi355.synth import rand_uniform, synth_state_dict
    from torchcfm_compat import SuperResModelWrapper

    dev = "cuda:0"
    net = SuperResModelWrapper(dim=(1, 28, 28), num_channels=32, num_res_blocks=1, num_classes=None, class_cond=True, precision="bf16")
    net.load_state_dict(synth_state_dict(param_shapes(net), 77))
    net.to(dev)
    imgs = rand_uniform(78, -1, 1, a.batch, 1, 28, 28).to(dev)
    torch.manual_seed(0)
    traj, low, nfe = utils_mnist_hy2.generate_samples_eval(net, imgs, batch_size=a.batch, solver=a.solver, steps=a.steps)
    torch.cuda.synchronize()
    print(f"solve done: traj {tuple(traj.shape)}, low_res {tuple(low.shape)}, network evaluations {nfe}, finite {bool(torch.isfinite(traj).all())}")


if __name__ == "__main__":
    main()
