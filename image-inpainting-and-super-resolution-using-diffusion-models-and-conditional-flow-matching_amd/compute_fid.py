"""Drop-in for the sampler half of the reference's `cifar10/compute_fid.py` on the MI355X backend.

`make_gen_1_img` builds the `gen_1_img(unused_latent) -> uint8 [B,3,32,32]` closure that cleanfid's
`fid.compute_fid(gen=...)` calls (cifar10/compute_fid.py:73-88), with the same flag names
(`integration_steps`, `integration_method`, `batch_size_fid`, `num_channel`).  With world_size > 1 every rank
samples its shard and the shards are collected with ONE RCCL all-gather (mi355.dist).  cleanfid itself
(Inception weights + CIFAR statistics, both downloaded) is not available offline; `main()` reports that.
"""
import argparse
import os

import torch

from mi355 import dist as mdist
from torchcfm_compat import UNetModelWrapper


def build_model(num_channel=128, device="cuda:0", precision=None):
    """cifar10/compute_fid.py:39-48."""
    return UNetModelWrapper(dim=(3, 32, 32), num_res_blocks=2, num_channels=num_channel, channel_mult=[1, 2, 2, 2], num_heads=4,
                            num_head_channels=64, attention_resolutions="16", dropout=0.1, precision=precision).to(device)


def load_checkpoint(net, path):
    """cifar10/compute_fid.py:52-65: take ["ema_model"], strip a 7-char "module." prefix on mismatch."""
    checkpoint = torch.load(path, map_location="cpu", weights_only=True)
    state_dict = checkpoint["ema_model"]
    try:
        net.load_state_dict(state_dict)
    except RuntimeError:
        net.load_state_dict({k[7:]: v for k, v in state_dict.items()})
    net.eval()
    return net


def draw_x0_shard(batch, seed, call_idx, device, shape=(3, 32, 32)):
    """This rank's slice of the batch's initial noise.  Every rank draws the SAME [batch, *shape] tensor from a generator seeded
    with seed + call_idx and keeps rows shard_range(batch), so an N-rank run integrates exactly the x0 of the 1-rank run
    (the reference's single `torch.randn(batch_size_fid, 3, 32, 32)`, cifar10/compute_fid.py:75, re-sharded)."""
    lo, hi = mdist.shard_range(batch)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + int(call_idx))
    return torch.randn(batch, *shape, device=device, generator=g)[lo:hi].contiguous()


def make_gen_1_img(new_net, batch_size_fid=1024, integration_steps=100, integration_method="euler", device="cuda:0", tol=1e-5, seed=0):
    if integration_method not in ("euler", "dopri5"):
        raise NotImplementedError("--integration_method must be euler or dopri5")
    device = torch.device(device)
    calls = [0]

    def gen_1_img(unused_latent):
        with torch.no_grad():
            B = int(batch_size_fid)
            x = draw_x0_shard(B, seed, calls[0], device)
            calls[0] += 1
            if integration_method == "euler":
                t_span = torch.linspace(0, 1, integration_steps + 1).tolist()
                _, _, img = new_net.engine(device).cfm_euler(x, t_span, want_u8=True)  # (traj*127.5+128).clip(0,255).to(uint8)
            else:  # odeint(new_net, x, linspace(0,1,2), rtol=tol, atol=tol, method="dopri5")  (cifar10/compute_fid.py:80-85)
                from mi355.ode import odeint_dopri5
                from mi355.ops import default_ops

                traj, _ = odeint_dopri5(lambda t, y: new_net(torch.tensor(float(t), device=device), y), x, 0.0, 1.0, tol, tol)
                img = default_ops.quantize_u8(traj.contiguous())
            return mdist.all_gather_batch(img, B)

    return gen_1_img


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_channel", type=int, default=128)
    ap.add_argument("--input_dir", default="./results")
    ap.add_argument("--model", default="otcfm")
    ap.add_argument("--integration_steps", type=int, default=100)
    ap.add_argument("--integration_method", default="dopri5")
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--step", type=int, default=400000)
    ap.add_argument("--num_gen", type=int, default=50000)
    ap.add_argument("--batch_size_fid", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0, help="x0 stream seed, shared by all ranks (each takes its batch slice)")
    a = ap.parse_args(argv)
    rank, world, local = mdist.init_from_env()
    device = f"cuda:{local}"
    net = build_model(a.num_channel, device)
    path = f"{a.input_dir}/{a.model}/{a.model}_cifar10_weights_step_{a.step}.pt"
    print("path: ", path)
    load_checkpoint(net, path)
    gen = make_gen_1_img(net, a.batch_size_fid, a.integration_steps, a.integration_method, device, a.tol, a.seed)
    try:
        from cleanfid import fid
    except ImportError as e:
        raise SystemExit("cleanfid is not installed (it downloads Inception weights and CIFAR statistics); "
                         "gen_1_img is ready for fid.compute_fid(gen=gen_1_img, ...) when it is") from e
    score = fid.compute_fid(gen=gen, dataset_name="cifar10", batch_size=a.batch_size_fid, dataset_res=32, num_gen=a.num_gen,
                            dataset_split="train", mode="legacy_tensorflow")
    if rank == 0:
        print("FID: ", score)


if __name__ == "__main__":
    main()
