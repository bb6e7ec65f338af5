#!/usr/bin/env python3
"""bench.py - sampled images/sec of the MI355X-native CFM sampler (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch: 50-step Euler CFM sampling of 256
CIFAR-10-shaped images (BASELINE.json configs[1]: cifar10/compute_fid.py --integration_method euler
--integration_steps 50, U-Net of cifar10/train_cifar10.py:92-101), bf16 contraction path, including the
final uint8 quantise and (N > 1) the single RCCL all-gather of the shards.  Inputs (x0, weights) are
resident in HBM before the timed region.  Weak scaling: every rank samples its own 256 images.

    python bench.py --gpus N --steps K --warmup W      (N > 1: launched under torch.distributed.run)

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = conv3x3_ws_kernel, measured live with
HIP events around every one of its launches in one forward, on the launch stream) and `cpu_baseline` (the fp32
PyTorch-CPU oracle timed on the host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (model kwargs for torchcfm-style wrapper, batch per GPU, Euler steps)
    "cifar10_cfm_euler50_b256": dict(dim=(3, 32, 32), num_res_blocks=2, num_channels=128, channel_mult=[1, 2, 2, 2], num_heads=4,
                                     num_head_channels=64, attention_resolutions="16", dropout=0.1),
}


def cpu_baseline(sd, steps_sample=10, batch=64, nfe=50):
    """fp32 PyTorch-CPU oracle (oracle/unet_ref.py + oracle/cfm_ref.py) on the host cores, bounded sample."""
    from oracle import cfm_ref, unet_ref

    cores = min(os.cpu_count() or 1, 16)  # a 1-GPU box grants a 16-core CPU share; more threads only oversubscribe it
    torch.set_num_threads(cores)
    cfg = unet_ref.UNetConfig(32, 3, 128, 3, 2, (2,), channel_mult=(1, 2, 2, 2), num_heads=4, num_head_channels=64)
    f = unet_ref.model_fn(sd, cfg)
    x = torch.randn(batch, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    ts = torch.linspace(0, 1, nfe + 1)[: steps_sample + 1]
    cfm_ref.euler_trajectory(f, x[:2], ts[:2], keep_all=False)  # warm-up
    t0 = time.perf_counter()
    cfm_ref.to_uint8(cfm_ref.euler_trajectory(f, x, ts, keep_all=False))
    dt = time.perf_counter() - t0
    ips = batch / (dt * nfe / steps_sample)
    return {"value": round(ips, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (fp32 PyTorch-CPU restatement) batch {batch}, {steps_sample} of {nfe} Euler steps in {dt:.2f} s, "
                      f"extrapolated to {nfe} steps; torch.set_num_threads({cores})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--nfe", type=int, default=50, help="Euler steps per sample")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-out", default="", help="write the per-op HIP-event profile of one forward to this JSON file")
    a = ap.parse_args()

    from mi355 import dist as mdist
    from mi355.synth import synth_state_dict

    rank, world, local = mdist.init_from_env()
    if world != a.gpus:
        if a.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from image_diffusion.unet import param_shapes
    from torchcfm_compat import UNetModelWrapper

    name = "cifar10_cfm_euler50_b256"
    net = UNetModelWrapper(precision=a.precision, **WORKLOADS[name])
    sd = synth_state_dict(param_shapes(net), 1234)  # no trained checkpoint exists offline: seeded, de-zeroed weights
    net.load_state_dict(sd)
    net.to(dev)
    eng = net.engine(dev)
    B, nfe = a.batch, a.nfe
    t_span = torch.linspace(0, 1, nfe + 1).tolist()
    g = torch.Generator(device=dev).manual_seed(rank)
    x0 = torch.randn(B, 3, 32, 32, device=dev, generator=g)

    def one_step():
        x = x0.clone()
        _, _, u8 = eng.cfm_euler(x, t_span, want_u8=True)
        return mdist.all_gather_batch(u8, B * world) if world > 1 else u8

    for _ in range(a.warmup):
        one_step()
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = one_step()
    torch.cuda.synchronize()
    mdist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape[0] == B * world and out.dtype == torch.uint8

    # ---- roofline of the dominant kernel, HIP events around every launch of one forward (on the launch stream) ----
    # Dominant kernel (rocprofv3 --kernel-trace: ~43 % of GPU time, profiles/): conv3x3_ws_kernel, the warp-specialised persistent
    # 3x3 implicit-GEMM; mi355_unet_profile reports its launches as tile_m == 256.  Algorithmic FLOPs = 2 * MACs of the conv.
    tt = torch.full((B,), 0.5, device=dev)
    eng.profile(x0, tt)  # warm
    recs = eng.profile(x0, tt)
    conv = [r for r in recs if r["kind"] == "conv"]
    dom = [r for r in conv if r["tile"][0] == 256] or conv
    by = {}
    for r in recs:
        d = by.setdefault(r["kind"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
        d["ms"] += r["ms"]; d["flops"] += r["flops"]; d["bytes"] += r["bytes"]; d["n"] += 1
    fwd_ms = sum(r["ms"] for r in recs)
    cms, cfl = by["conv"]["ms"], by["conv"]["flops"]
    dms, dfl, dby = sum(r["ms"] for r in dom), sum(r["flops"] for r in dom), sum(r["bytes"] for r in dom)
    peak = PEAK_BF16_TFLOPS if a.precision == "bf16" else PEAK_F32_TFLOPS
    achieved = dfl / (dms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    pmc_file = os.path.join(REPO, "profiles", "r1_pmc_hbm_traffic.json")   # rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py)
    if os.path.exists(pmc_file) and a.precision == "bf16" and B == 256:
        pj = json.load(open(pmc_file))
        ent = pj.get("kernels", {}).get("conv3x3_ws_kernel")
        if ent:
            traffic, traffic_src = ent["hbm_bytes_per_launch"], "profiles/r1_pmc_hbm_traffic.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes)"
    roofline = {
        "kernel": "conv3x3_ws_kernel<%s> (warp-specialised persistent 3x3 implicit-GEMM; %d of %d conv launches per forward)" % (a.precision, len(dom), len(conv)),
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_us": round(1e3 * dms / len(dom), 2),
        "flops_per_launch": dfl / len(dom),
        "algorithmic_bytes_per_launch": dby / len(dom),
        "share_of_forward": round(dms / fwd_ms, 3),
        "all_conv_kernels": {"launches": len(conv), "achieved_tflops": round(cfl / (cms * 1e-3) / 1e12, 2), "share_of_forward": round(cms / fwd_ms, 3)},
        "forward_ms_by_kind": {k: round(v["ms"], 3) for k, v in by.items()},
    }
    if a.profile_out and rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.profile_out)), exist_ok=True)
        json.dump({"batch": B, "precision": a.precision, "ops": recs}, open(a.profile_out, "w"), indent=1)

    res = {
        "metric": "sampled images/sec (50-step, 32x32)", "value": round(B * world * a.steps / dt, 2), "unit": "images/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": name, "images_per_gpu_per_step": B, "euler_steps": nfe, "image": "3x32x32",
                   "unet": "mc128 mult(1,2,2,2) 2 resblocks attn@16x16 heads4x64 (35.7M params)", "parallelism": f"dp{world} batch-sharded",
                   "weights": "synthetic seeded (no checkpoint offline)"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(sd)
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
