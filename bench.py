#!/usr/bin/env python3
"""bench.py - sampled images/sec of the MI355X-native sampler (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch.  The default workload is BASELINE.json configs[1]:
50-step Euler CFM sampling of 256 CIFAR-10-shaped images (cifar10/compute_fid.py --integration_method euler
--integration_steps 50, U-Net of cifar10/train_cifar10.py:92-101), bf16 contraction path, including the final uint8
quantise and (N > 1) the single RCCL all-gather of the shards.  Inputs (x0, condition, weights) are resident in HBM
before the timed region.  Weak scaling: every rank samples its own batch.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--precision bf16|bf16x2|fp16|fp32]
    N > 1: either launched under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or plain
    `python bench.py --gpus N ...`: the parent then starts N fresh children of itself, one per GPU, with that environment (it has
    not touched the GPU at that point), relays rank 0's JSON line and exits with the worst child's code.
    --dry-run: stop before the first GPU call; every rank joins a gloo group, checks its environment and its batch shard, rank 0
    prints one JSON line (the launcher's CPU test, tests/test_host_logic.py).

Other workloads (--workload; the parity tests of the same configurations are tests/test_gpu_configs.py):
    cifar10_inpaint_ddpm50_b512        BASELINE cfg 3: CIFAR U-Net in=6, centred 16x16 = -2, Amortized DDPM Ns=50, device Philox
    cifar10_inpaint_ddim50_b512        the same with the build-defined DDIM(eta=0) extension (cfg 3 names DDIM)
    flowers64_superres_euler100_b256   BASELINE cfg 4 shard: Flowers-64 net (FiLM, up/down ResBlocks) in=6 = x || bilinear(16->64), 100 Euler
    px128_inpaint_ddim100_b128         BASELINE cfg 5 shard: 128 px, attention at 32/16/8, in=6, free-form mask, DDIM Ns=100
    cifar64_cfm_euler50_b256           64x64 unconditional (north_star: "32x32 and 64x64 batches"): CIFAR net widened to 64 px

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, measured live with HIP events around every one of its
launches in one forward, on the launch stream; plus the whole-path MFMA / HBM fractions) and `cpu_baseline` (the fp32
PyTorch-CPU oracle timed on the host cores on a bounded sample; rank 0, N = 1, default workload only).
"""
import argparse
import ctypes as C
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
BOX_REF_US = 475.0   # reference duration of the frozen box probe launch (csrc/box_probe.hip): lines are compared as value * launch_us / BOX_REF_US

CIFAR = dict(image_size=32, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2, attention_resolutions=(2,),
             channel_mult=(1, 2, 2, 2), num_heads=4, num_head_channels=64)
FLOWERS = dict(image_size=64, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=1, attention_resolutions=(4,),
               channel_mult=(1, 2, 3, 4), num_heads=4, num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True)
PX128 = dict(image_size=128, in_channels=6, model_channels=128, out_channels=3, num_res_blocks=1, attention_resolutions=(4, 8, 16),
             channel_mult=(1, 1, 2, 3, 4), num_heads=4, num_head_channels=64)

WORKLOADS = {
    # name: net = UNetModel kwargs, kind = cfm | ddpm | ddim, batch per GPU, nfe, cond = None | center | superres | freeform
    "cifar10_cfm_euler50_b256": dict(net=CIFAR, kind="cfm", batch=256, nfe=50, cond=None,
                                     metric="sampled images/sec (50-step, 32x32)",
                                     unet="mc128 mult(1,2,2,2) 2 resblocks attn@16x16 heads4x64 (35.7M params)"),
    "cifar10_inpaint_ddpm50_b512": dict(net=dict(CIFAR, in_channels=6), kind="ddpm", batch=512, nfe=50, cond="center",
                                        metric="sampled images/sec (50-step Amortized DDPM, 32x32 centre-mask in-painting)",
                                        unet="CIFAR arch, in_channels 6 (x || condition)"),
    "cifar10_inpaint_ddim50_b512": dict(net=dict(CIFAR, in_channels=6), kind="ddim", batch=512, nfe=50, cond="center",
                                        metric="sampled images/sec (50-step DDIM, 32x32 centre-mask in-painting)",
                                        unet="CIFAR arch, in_channels 6 (x || condition)"),
    "flowers64_superres_euler100_b256": dict(net=FLOWERS, kind="cfm", batch=256, nfe=100, cond="superres",
                                             metric="sampled images/sec (100-step Euler, 64x64 4x super-resolution)",
                                             unet="Flowers-64: mc128 mult(1,2,3,4) FiLM up/down ResBlocks attn@16x16, in 6 (68.2M params)"),
    "px128_inpaint_ddim100_b128": dict(net=PX128, kind="ddim", batch=128, nfe=100, cond="freeform",
                                       metric="sampled images/sec (100-step DDIM, 128x128 free-form in-painting)",
                                       unet="128 px: mc128 mult(1,1,2,3,4) attn@32/16/8 heads x64, in 6 (74.6M params)"),
    "cifar64_cfm_euler50_b256": dict(net=dict(CIFAR, image_size=64, attention_resolutions=(4,)), kind="cfm", batch=256, nfe=50, cond=None,
                                     metric="sampled images/sec (50-step, 64x64)",
                                     unet="CIFAR arch at 64x64 (attention at 16x16)"),
}
DEFAULT = "cifar10_cfm_euler50_b256"


def cpu_baseline(sd, steps_sample=10, batch=32, nfe=50, reps=2):
    """fp32 PyTorch-CPU oracle (oracle/unet_ref.py + oracle/cfm_ref.py) on the host cores, bounded sample (batch 32,
    >= 2 repetitions).  Each repetition runs `steps_sample` of the `nfe` Euler steps on a fresh batch; every step costs the same
    (one network evaluation + an axpy), so the rate is scaled to nfe steps.  `value` is the mean of the repetitions."""
    from oracle import cfm_ref, unet_ref

    cores = min(os.cpu_count() or 1, 16)  # a 1-GPU box grants a 16-core CPU share; more threads only oversubscribe it
    torch.set_num_threads(cores)
    cfg = unet_ref.UNetConfig(32, 3, 128, 3, 2, (2,), channel_mult=(1, 2, 2, 2), num_heads=4, num_head_channels=64)
    f = unet_ref.model_fn(sd, cfg)
    ts = torch.linspace(0, 1, nfe + 1)[: steps_sample + 1]
    gen = torch.Generator().manual_seed(0)
    cfm_ref.euler_trajectory(f, torch.randn(2, 3, 32, 32, generator=gen), ts[:2], keep_all=False)  # warm-up
    secs = []
    for _ in range(reps):
        x = torch.randn(batch, 3, 32, 32, generator=gen)
        t0 = time.perf_counter()
        cfm_ref.to_uint8(cfm_ref.euler_trajectory(f, x, ts, keep_all=False))
        secs.append(time.perf_counter() - t0)
    rates = [batch / (dt * nfe / steps_sample) for dt in secs]
    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(sum(rates) / len(rates), 4), "unit": "images/s", "cores": cores, "kind": "port",
            "repetitions": [round(r, 4) for r in rates], "cpu": cpu_model,
            "sample": f"oracle (fp32 PyTorch-CPU restatement) batch {batch}, {reps} repetitions of {steps_sample} of {nfe} Euler steps "
                      f"({', '.join('%.2f s' % d for d in secs)}), each scaled to {nfe} steps; torch.set_num_threads({cores})"}


def make_condition(kind, B, C, S, dev, seed):
    """Synthetic condition tensors of SURVEY 8(d): U(-1,1) images with the masked pixels set to the -2 sentinel, or a U(-1,1)
    low-res image bilinearly upsampled (built once, outside the timed region: the reference builds them once per batch too)."""
    if kind is None:
        return None
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    if kind == "superres":
        import torch.nn.functional as F

        low = torch.rand(B, C, S // 4, S // 4, generator=g) * 2 - 1
        return F.interpolate(low, (S, S), mode="bilinear").to(dev).contiguous()
    cond = torch.rand(B, C, S, S, generator=g) * 2 - 1
    if kind == "center":
        cond[:, :, S // 4: 3 * S // 4, S // 4: 3 * S // 4] = -2.0
    elif kind == "freeform":
        from mi355.synth import free_form_mask

        m = free_form_mask(2000 + seed, min(B, 16), S, S, 0.4)
        m = m.repeat((B + m.shape[0] - 1) // m.shape[0], 1, 1, 1)[:B]
        cond = torch.where(m.expand_as(cond), torch.full_like(cond, -2.0), cond)
    return cond.to(dev).contiguous()


def self_launch(n, port=0):
    """`python bench.py --gpus N` without a launcher: start N children of this script, one per GPU, with the environment
    torch.distributed.run would give them (rendezvous on 127.0.0.1), relay what they print, return the worst exit code.
    The parent never initialises the GPU: it only spawns and waits (children are fresh interpreters, not forks or execs)."""
    import socket
    import subprocess

    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    import tempfile
    import time

    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL across processes)
        # every rank's output goes to a file of its own (a pipe nobody drains can block a child; rank > 0 output is kept for failures)
        logs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=logs[-1], stderr=subprocess.STDOUT if r else None))
    # Poll ALL children: if any rank dies early (import error, GPU fault, out of memory) the others would sit in the rendezvous or in a
    # collective until the process-group timeout; on the first non-zero exit the siblings are terminated (they are plain children of this
    # process: terminate / kill by handle, nothing is matched by pattern) and the failure is reported with the failed ranks' output.
    rcs = [None] * n
    failed, first_bad = False, 0
    while any(rc is None for rc in rcs):
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()
                if rcs[r] not in (None, 0):
                    failed, first_bad = True, first_bad or rcs[r]
        if failed:
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    pr.terminate()
            t_end = time.time() + 10
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = pr.wait(timeout=max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        rcs[r] = pr.wait()
            break
        time.sleep(0.05)
    logs[0].seek(0)
    out0 = logs[0].read()
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    worst = first_bad or max(rcs, key=lambda c: (c != 0, abs(c)))   # the rank that failed by itself, not the siblings this parent then terminated
    if worst:
        sys.stderr.write(f"bench.py: child exit codes {rcs}\n")
        for r in range(1, n):
            logs[r].seek(0)
            tail = logs[r].read()[-2000:]
            if tail.strip():
                sys.stderr.write(f"---- rank {r} output (tail) ----\n{tail}\n")
    for f in logs:
        f.close()
    return worst


def dry_run(a, mdist):
    """The N > 1 path up to the first GPU call: environment, process group (gloo), this rank's shard of the global batch."""
    if os.environ.get("MI355_BENCH_TEST_DIE_RANK", "") == os.environ.get("RANK", "?"):   # tests only: a rank that dies before the rendezvous
        sys.stderr.write("dying on request\n")
        return 7
    rank, world, local = mdist.init_from_env(backend="gloo")
    ok = world == a.gpus and 0 <= rank < world and local == rank
    B = (a.batch or WORKLOADS[a.workload]["batch"])
    lo, hi = mdist.shard_range(B * world, rank, world)   # weak scaling: B images per rank of a global batch B * world
    ok = ok and (hi - lo) == B and lo == rank * B
    if world > 1:
        t = torch.tensor([rank, lo, hi], dtype=torch.int64)
        g = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        torch.distributed.all_gather(g, t)
        shards = [[int(v) for v in x] for x in g]
        flag = torch.tensor([0 if ok else 1])
        torch.distributed.all_reduce(flag)
        ok = int(flag.item()) == 0 and [s[0] for s in shards] == list(range(world)) and all(s[1] == k * B for k, s in enumerate(shards))
        mdist.barrier()
    else:
        shards = [[rank, lo, hi]]
    if rank == 0:
        print(json.dumps({"dry_run": True, "ok": bool(ok), "n_gpus": world, "shards": shards, "master": os.environ.get("MASTER_ADDR"),
                          "workload": a.workload}))
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT, choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the workload's)")
    ap.add_argument("--nfe", type=int, default=0, help="network evaluations per sample (default: the workload's)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x2", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-out", default="", help="write the per-op HIP-event profile of one forward to this JSON file")
    ap.add_argument("--dry-run", action="store_true", help="launcher check: process group + shard ranges over gloo, no GPU call")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (default: a free one)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus, a.master_port))   # nothing above has touched the GPU (no HIP call, no library load)

    from mi355 import dist as mdist

    if a.dry_run:
        raise SystemExit(dry_run(a, mdist))

    from mi355 import _lib
    from mi355.ops import default_ops
    from mi355.synth import synth_state_dict

    rank, world, local = mdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from image_diffusion.sde_diffusion import DDPM
    from image_diffusion.unet import UNetModel, param_shapes

    name = a.workload
    wl = WORKLOADS[name]
    net = UNetModel(precision=a.precision, **wl["net"])
    sd = synth_state_dict(param_shapes(net), 1234)  # no trained checkpoint exists offline: seeded, de-zeroed weights
    net.load_state_dict(sd)
    net.to(dev)
    eng = net.engine(dev)
    B, nfe = a.batch or wl["batch"], a.nfe or wl["nfe"]
    S, Cx = wl["net"]["image_size"], wl["net"]["out_channels"]
    g = torch.Generator(device=dev).manual_seed(rank)
    x0 = torch.randn(B, Cx, S, S, device=dev, generator=g)
    cond = make_condition(wl["cond"], B, Cx, S, dev, rank)
    t_span = torch.linspace(0, 1, nfe + 1).tolist()
    tables = DDPM(nfe).host_tables() if wl["kind"] != "cfm" else None
    step_no = [0]

    def one_step():
        x = x0.clone()
        if wl["kind"] == "cfm":
            _, _, u8 = eng.cfm_euler(x, t_span, cond=cond, want_u8=True)
        else:
            step_no[0] += 1
            mode = _lib.DDPM_AMORTIZED if wl["kind"] == "ddpm" else _lib.DDIM
            eng.ddpm_sample(x, tables, mode=mode, cond=cond, seed=(rank << 32) + step_no[0])   # device Philox noise
            u8 = default_ops.quantize_u8(x)
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

    # ---- box calibration, right behind the timed region (same thermal / clock state): a FROZEN MFMA-only launch (csrc/box_probe.hip) ----
    box = None
    if rank == 0:
        L = _lib.lib()
        pw = torch.empty(int(L.mi355_box_probe_workspace_bytes()), device=dev, dtype=torch.uint8)
        us, mhz, tfl = C.c_float(), C.c_float(), C.c_float()
        _lib.check(L.mi355_box_probe(20, C.c_void_p(pw.data_ptr()), pw.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                     C.byref(us), C.byref(mhz), C.byref(tfl)), "mi355_box_probe")
        box = {"kernel": "box_probe_kernel (frozen: 1.0995 TFLOP of bf16 16x16x32 MFMA on seeded register operands, no memory traffic)",
               "launch_us": round(us.value, 2), "tflops": round(tfl.value / (us.value * 1e-6), 1), "in_kernel_clock_mhz": round(mhz.value, 0),
               "reference_launch_us": BOX_REF_US,
               "note": "boxes of the pool differ by several per cent for one build: compare lines of different boxes / rounds as value * launch_us / reference_launch_us"}
        box["value_at_reference_box"] = round(B * world * a.steps / dt * us.value / BOX_REF_US, 2)
        # the memory side of the same calibration: a FROZEN 512-MiB -> 512-MiB copy (csrc/box_probe_hbm.hip); its buffer is freed before the profile pass
        hw = torch.empty(int(L.mi355_box_probe_hbm_workspace_bytes()), device=dev, dtype=torch.uint8)
        hus, hgb = C.c_float(), C.c_float()
        _lib.check(L.mi355_box_probe_hbm(10, C.c_void_p(hw.data_ptr()), hw.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                         C.byref(hus), C.byref(hgb)), "mi355_box_probe_hbm")
        del hw
        box["hbm_copy"] = {"kernel": "box_probe_hbm_kernel (frozen: 512 MiB read + 512 MiB written per launch, 16-byte accesses)",
                           "launch_us": round(hus.value, 2), "gbs": round(hgb.value / (hus.value * 1e-6), 1)}

    # ---- roofline of the dominant kernel, HIP events around every launch of one forward (on the launch stream) ----
    # Dominant kernel (rocprofv3 --kernel-trace, profiles/): conv3x3_ws_kernel, the warp-specialised persistent 3x3 implicit-GEMM;
    # mi355_unet_profile reports its launches as tile_m == 256.  Algorithmic FLOPs = 2 * MACs of the conv.
    tt = torch.full((B,), 0.5, device=dev)
    eng.profile(x0, tt, cond)  # warm
    recs_all = eng.profile(x0, tt, cond)
    recs = [r for r in recs_all if r["tile"][0] >= 0]   # tile (-1, -1): a plan op that launched nothing (fused into a neighbour)
    conv = [r for r in recs if r["kind"] == "conv"]
    k3 = [r for r in conv if r["ks"] == 3 and r["tile"][0] in (256, 512)]
    dom = [r for r in k3 if tuple(r["tile"]) == (256, 128)] or k3 or conv    # conv3x3_ws_kernel: 256 px x 128 ch tiles
    ppk = [r for r in k3 if tuple(r["tile"]) in ((256, 256), (512, 128))]    # conv3x3_pp_kernel: 256 px x 256 ch (wide) / 512 px x 128 ch (narrow) tiles
    k1 = [r for r in conv if r["ks"] == 1]
    by = {}
    for r in recs:
        d = by.setdefault(r["kind"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
        d["ms"] += r["ms"]; d["flops"] += r["flops"]; d["bytes"] += r["bytes"]; d["n"] += 1
    fwd_ms = sum(r["ms"] for r in recs)
    cms, cfl = by["conv"]["ms"], by["conv"]["flops"]
    dms, dfl, dby = sum(r["ms"] for r in dom), sum(r["flops"] for r in dom), sum(r["bytes"] for r in dom)
    peak = PEAK_BF16_TFLOPS if a.precision in ("bf16", "bf16x2", "fp16") else PEAK_F32_TFLOPS
    achieved = dfl / (dms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    for pmc_name in ("r5_pmc_hbm_traffic.json", "r4_pmc_hbm_traffic.json", "r3_pmc_hbm_traffic.json", "r2_pmc_hbm_traffic.json", "r1_pmc_hbm_traffic.json"):   # rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py)
        pmc_file = os.path.join(REPO, "profiles", pmc_name)
        if os.path.exists(pmc_file) and a.precision == "bf16" and name == DEFAULT and B == 256:
            ent = json.load(open(pmc_file)).get("kernels", {}).get("conv3x3_ws_kernel")
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{pmc_name} (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes)"
                break
    def leg(rs, kname, bound):
        """roofline entry of one more kernel family of the same forward (same accounting as the dominant kernel's)"""
        if not rs:
            return None
        ms, fl, by_ = sum(r["ms"] for r in rs), sum(r["flops"] for r in rs), sum(r["bytes"] for r in rs)
        e = {"kernel": kname, "launches": len(rs), "avg_launch_us": round(1e3 * ms / len(rs), 2), "share_of_forward": round(ms / fwd_ms, 3),
             "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 2), "achieved_gbs": round(by_ / (ms * 1e-3) / 1e9, 1), "bound": bound}
        e["frac"] = round(e["achieved_tflops"] / peak, 4) if bound == "mfma" else round(e["achieved_gbs"] / PEAK_HBM_GBS, 4)
        return e
    other = {"conv3x3_pp_kernel": leg(ppk, "conv3x3_pp_kernel<%s> (ping-pong 3x3, MFMA waves issue their own LDS-DMA behind counted vmcnt)" % a.precision, "mfma"),
             "conv1x1": leg(k1, "conv1x1_pp_kernel / conv1x1 kernels (all 1x1 convs of the forward)", "hbm"),
             "attention_block": leg([r for r in recs if r["kind"] == "attention" and r["ks"] == 1],
                                    "attn_fused_pers_kernel / attn_fused_kernel (GroupNorm-apply + qkv + attention in one launch)", "mfma")}
    # whole path: algorithmic FLOPs / bytes of one network evaluation (SURVEY 8d accounting: the engine's plan counts 2*MAC of every
    # contraction and in + out activation bytes of every contraction op, weights once) over the measured time per evaluation
    st = eng.stats(B)
    per_eval_s = dt / (a.steps * nfe)
    flops_eval = st["conv_flops"] + st["attn_flops"]
    bytes_eval = st["act_bytes"] + st["weight_bytes"]
    roofline = {
        "kernel": "conv3x3_ws_kernel<%s> (warp-specialised persistent 3x3 implicit-GEMM; %d of %d conv launches per forward)" % (a.precision, len(dom), len(conv)),
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_us": round(1e3 * dms / len(dom), 2),
        "flops_per_launch": dfl / len(dom),
        "algorithmic_bytes_per_launch": dby / len(dom),
        "share_of_forward": round(dms / fwd_ms, 3),
        "box": box,
        "other_kernels": other,
        "all_conv_kernels": {"launches": len(conv), "achieved_tflops": round(cfl / (cms * 1e-3) / 1e12, 2), "share_of_forward": round(cms / fwd_ms, 3)},
        "forward_ms_by_kind": {k: round(v["ms"], 3) for k, v in by.items()},
        "whole_path": {
            "ms_per_network_evaluation": round(1e3 * per_eval_s, 4), "launches_per_evaluation": st["launches"],
            "algorithmic_tflop_per_evaluation": round(flops_eval / 1e12, 4), "algorithmic_gb_per_evaluation": round(bytes_eval / 1e9, 4),
            "achieved_mfma": round(flops_eval / per_eval_s / 1e12, 1), "achieved_mfma_frac": round(flops_eval / per_eval_s / 1e12 / peak, 4),
            "achieved_hbm": round(bytes_eval / per_eval_s / 1e9, 1), "achieved_hbm_frac": round(bytes_eval / per_eval_s / 1e9 / PEAK_HBM_GBS, 4),
            "note": "time per evaluation = timed region / (steps * nfe): includes the step-update / quantise kernels and (N>1) the all-gather",
        },
    }
    if a.profile_out and rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.profile_out)), exist_ok=True)
        json.dump({"workload": name, "batch": B, "precision": a.precision, "ops": recs_all}, open(a.profile_out, "w"), indent=1)

    res = {
        "metric": wl["metric"], "value": round(B * world * a.steps / dt, 2), "unit": "images/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": name, "images_per_gpu_per_step": B, "network_evaluations_per_sample": nfe, "sampler": wl["kind"],
                   "image": f"{Cx}x{S}x{S}", "condition": wl["cond"], "unet": wl["unet"], "parallelism": f"dp{world} batch-sharded",
                   "weights": "synthetic seeded (no checkpoint offline)"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline and name == DEFAULT:
        res["cpu_baseline"] = cpu_baseline(sd)
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
