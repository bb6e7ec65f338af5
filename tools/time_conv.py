#!/usr/bin/env python3
"""Time single conv shapes (GPU events around repeated launches of the test op's conv kernel via rocprof-free timing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.conftest  # noqa
import torch
from mi355.ops import default_ops as ops
from mi355 import _lib
from mi355.synth import synth_state_dict
dev = "cuda:0"
B, Cin, H, Cout, k = [int(v) for v in sys.argv[1:6]]
NOGN = len(sys.argv) > 6 and sys.argv[6] == "nogn"   # conv without GN / SiLU prologue (the small levels after gn_affine's apply pass)
x = torch.randn(B, Cin, H, H, device=dev)
sd = synth_state_dict({"g": (Cin,), "b": (Cin,), "weight": (Cout, Cin, k, k), "bias": (Cout,)}, 1)
gnp = (sd["g"].to(dev), sd["b"].to(dev))
for _ in range(2):
    ops.conv2d(x, sd["weight"], sd["bias"], gn=None if NOGN else gnp, gn_silu=not NOGN, dtype=_lib.MI355_BF16)
torch.cuda.synchronize()
print("ok")
