import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.conftest  # noqa
import torch, torch.nn.functional as F
from mi355.ops import default_ops as ops
from mi355 import _lib
torch.set_printoptions(precision=3, linewidth=200, sci_mode=False)
dev = "cuda:0"
def run(B, Cin, H, Cout, k, dtype, ident=False):
    x = torch.arange(B*Cin*H*H, dtype=torch.float32).reshape(B, Cin, H, H) % 7 - 3
    w = torch.zeros(Cout, Cin, k, k)
    if ident:
        for c in range(min(Cin, Cout)): w[c, c, k//2, k//2] = 1.0
    else:
        w = ((torch.arange(w.numel(), dtype=torch.float32).reshape(w.shape) % 5) - 2) * 0.25
    b = torch.zeros(Cout)
    ref = F.conv2d(x, w, b, padding=k//2)
    got = ops.conv2d(x.to(dev), w, b, dtype=dtype).cpu()
    err = (got - ref).abs()
    print(f"B{B} Cin{Cin} H{H} Cout{Cout} k{k} dtype{dtype} ident{ident}: max err {err.max():.4f}")
    if err.max() > 1e-2:
        bad = (err > 1e-2).nonzero()
        print(" n bad", len(bad), "of", err.numel(), "first", bad[:5].tolist())
        print(" got[0,:4,:3,:6]\n", got[0, :4, :3, :6], "\n ref\n", ref[0, :4, :3, :6])
for dt in (0, 1):
    run(1, 32, 8, 32, 1, dt, True)
    run(1, 32, 8, 32, 1, dt, False)
    run(1, 32, 8, 32, 3, dt, True)
    run(2, 32, 8, 64, 3, dt, False)
