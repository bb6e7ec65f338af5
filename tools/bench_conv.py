#!/usr/bin/env python3
"""Micro-benchmark of single conv shapes through the test op mi355_conv2d (for rocprofv3 PMC runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.conftest  # noqa
import torch
from mi355.ops import default_ops as ops
from mi355 import _lib
from mi355.synth import randn, synth_state_dict
dev = "cuda:0"
shapes = [  # B, Cin, H, Cout, k, gn
    (256, 128, 32, 128, 3, True),
    (256, 256, 16, 256, 3, True),
    (256, 256, 16, 768, 1, True),
    (256, 256, 4, 256, 3, True),
]
if len(sys.argv) > 1:
    shapes = [shapes[int(a)] for a in sys.argv[1:]]
for (B, Cin, H, Cout, k, gn) in shapes:
    x = torch.randn(B, Cin, H, H, device=dev)
    sd = synth_state_dict({"g": (Cin,), "b": (Cin,), "weight": (Cout, Cin, k, k), "bias": (Cout,)}, 1)
    gnp = (sd["g"].to(dev), sd["b"].to(dev)) if gn else None
    for _ in range(3):
        y = ops.conv2d(x, sd["weight"], sd["bias"], gn=gnp, gn_silu=True, dtype=_lib.MI355_BF16)
    torch.cuda.synchronize()
    print("done", B, Cin, H, Cout, k, float(y.abs().mean()))
