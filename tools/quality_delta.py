#!/usr/bin/env python3
"""Quality delta of the measured (bf16) mode against the parity-green (fp32) mode: SURVEY.md 8(d)(ii), VERDICT r1 task 7.

BASELINE configs[1] (CIFAR U-Net, 50 Euler steps), N samples (default 10 240 = 40 batches of 256) integrated from IDENTICAL x0
in bf16 mode and in fp32 mode (fp32 mode is the one held to the CPU oracle at 2e-4: the oracle itself samples ~1 image/s and
cannot produce 10 k).  Reports
  * per-sample error of the final 50-step state: max-abs, RMS, and the uint8 code differences;
  * "FID-proxy delta": Frechet distance between the two uint8 sample sets in the seeded random-conv feature space of
    evaluation.random_conv_features, next to (a) the same-distribution floor = fp32 samples from two disjoint x0 sets and
    (b) a scale reference = fp32 samples with 25 instead of 50 Euler steps (a real change of the sampler).
True FID needs cleanfid's downloaded Inception weights + CIFAR statistics (cifar10/compute_fid.py:92-100): unavailable offline.

Round 3 (task 6): with N(0, 0.02^2) synthetic weights the learned-field stand-in barely bends the trajectories (50 vs
25 Euler steps differed LESS than bf16 vs fp32), so the proxy could not tell samplers apart.  `--gain auto` (the default now) rescales
the last conv (out.2) so that the field's rms at t = 0 is 1: |x1 - x0| becomes comparable to |x0|, the deep net's x-dependence bends
the paths, and a change of the step count becomes visible.  Reported side by side, all from the same x0: bf16 vs fp32, 50 vs 25
steps, 50 vs 49 steps, plus the disjoint-x0 floor; the claim to check is "bf16 moves the sample distribution less than dropping
ONE Euler step does".

Round 4 (task 5): `--split` asks WHICH roundings carry the bf16 mode's per-sample error.  The engine has one element type
per plan, so the split is made with the two modes it has and with weights / inputs that are rounded to bf16 BEFORE they are handed over:
  weights            fp32 mode on bf16-rounded conv weights vs fp32 mode on the original ones (all / 3x3 only / 1x1 + qkv + proj only /
                     first + last conv only): the weight roundings alone;
  everything else    bf16 mode vs fp32 mode, BOTH on the bf16-rounded weights (exactly representable: the bf16 engine packs them
                     without error): activation storage, bf16 MFMA operands, bf16 residual trunk - all that is not weight rounding;
  network input      fp32 mode with the state rounded to bf16 in front of every evaluation (host loop over mi355_unet_forward_t; the Euler
                     update itself stays fp32): the first conv's input quantisation alone.
Finer splits of "everything else" (trunk in fp32, fp32 only around conv 0 / out) need kernels with mixed element types, which this
engine does not have.

    python tools/quality_delta.py [--n 10240] [--gain auto|1.0|<float>] [--out gpurun_out/r3_quality_delta.json]
    python tools/quality_delta.py --split [--n 5120] --out profiles/r4_quality_delta.json
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10240)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--nfe", type=int, default=50)
    ap.add_argument("--gain", default="auto", help="multiplier of out.2 (weight and bias); auto = 1 / rms of the field at t = 0")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r3_quality_delta.json"))
    ap.add_argument("--split", action="store_true", help="round 4: which roundings carry the bf16 mode's error (see the module docstring)")
    ap.add_argument("--x2", action="store_true", help="round 5: bf16 and bf16x2 (hi + lo weight halves) against fp32 mode, same x0: per-sample rms of the final state")
    a = ap.parse_args()

    import evaluation
    from compute_fid import build_model
    from image_diffusion.unet import param_shapes
    from mi355.synth import synth_state_dict

    dev = torch.device("cuda:0")
    net = build_model(128, dev, precision="bf16")
    sd = synth_state_dict(param_shapes(net), 1234)
    net.load_state_dict(sd)
    nb = (a.n + a.batch - 1) // a.batch
    # field gain: rms of v(x0, t = 0) with the plain synthetic weights -> scale out.2 so that it becomes 1
    g0 = torch.Generator(device=dev).manual_seed(0)
    xprobe = torch.randn(a.batch, 3, 32, 32, device=dev, generator=g0)
    net.set_precision("fp32")
    v_rms_plain = float(net.engine(dev).forward(xprobe, 0.0).pow(2).mean().sqrt())
    gain = (1.0 / v_rms_plain) if a.gain == "auto" else float(a.gain)
    sd = dict(sd)
    sd["out.2.weight"] = sd["out.2.weight"] * gain
    sd["out.2.bias"] = sd["out.2.bias"] * gain
    net.load_state_dict(sd)
    v_rms = float(net.engine(dev).forward(xprobe, 0.0).pow(2).mean().sqrt())
    print(f"field rms at t=0: {v_rms_plain:.4f} with N(0, 0.02^2) weights -> {v_rms:.4f} with out.2 x {gain:.3f}", flush=True)

    moves = {}

    def sample_set(precision, seed0, nfe):
        net.set_precision(precision)
        eng = net.engine(dev)
        ts = torch.linspace(0, 1, nfe + 1).tolist()
        xs, u8s = [], []
        move = []
        t0 = time.perf_counter()
        for k in range(nb):
            g = torch.Generator(device=dev).manual_seed(seed0 + k)
            x = torch.randn(a.batch, 3, 32, 32, device=dev, generator=g)
            x0 = x.clone()
            _, _, u8 = eng.cfm_euler(x, ts, want_u8=True)
            move.append(float(((x - x0).pow(2).sum() / x0.pow(2).sum()).sqrt()))
            xs.append(x.cpu()); u8s.append(u8.cpu())
        moves[(precision, seed0, nfe)] = sum(move) / len(move)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{precision} nfe {nfe} seeds {seed0}..: {nb * a.batch} samples in {dt:.1f} s ({nb * a.batch / dt:.0f} img/s)", flush=True)
        return torch.cat(xs), torch.cat(u8s)

    if a.split:
        return split(a, net, sd, dev, nb, gain, v_rms)
    if a.x2:
        x32, _ = sample_set("fp32", 0, a.nfe)
        x16, _ = sample_set("bf16", 0, a.nfe)
        xx2, _ = sample_set("bf16x2", 0, a.nfe)
        xh, _ = sample_set("fp16", 0, a.nfe)
        x32m, _ = sample_set("fp32", 0, a.nfe - 1)
        rms = lambda d: float(d.pow(2).mean().sqrt())
        res = {"workload": "cifar10_cfm_euler50 (BASELINE configs[1] net, synthetic seeded weights)", "n_samples": int(x32.shape[0]), "nfe": a.nfe,
               "field": {"out2_gain": gain, "rms_v_t0": v_rms},
               "per_sample_rms_of_the_final_state_same_x0": {"bf16_vs_fp32": rms(x16 - x32), "bf16x2_vs_fp32": rms(xx2 - x32), "fp16_vs_fp32": rms(xh - x32),
                                                             f"scale: {a.nfe}_vs_{a.nfe - 1}_steps (fp32 mode)": rms(x32m - x32)},
               "max_abs": {"bf16_vs_fp32": float((x16 - x32).abs().max()), "bf16x2_vs_fp32": float((xx2 - x32).abs().max()), "fp16_vs_fp32": float((xh - x32).abs().max())}}
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
        print(json.dumps(res))
        return
    x16, u16 = sample_set("bf16", 0, a.nfe)
    x32, u32 = sample_set("fp32", 0, a.nfe)
    _, u32b = sample_set("fp32", 100000, a.nfe)          # disjoint x0: same-distribution floor
    x32h, u32h = sample_set("fp32", 0, a.nfe // 2)       # half the Euler steps: a real sampler change, for scale
    x32m, u32m = sample_set("fp32", 0, a.nfe - 1)        # ONE Euler step fewer: the smallest change of the sampler one can make
    err = (x16 - x32).abs()
    per_sample_rms = (x16 - x32).pow(2).mean(dim=(1, 2, 3)).sqrt()
    code = (u16.int() - u32.int()).abs()
    f16, f32, f32b, f32h, f32m = (evaluation.random_conv_features(u, seed=0) for u in (u16, u32, u32b, u32h, u32m))
    rms = lambda d: float(d.pow(2).mean().sqrt())
    res = {
        "workload": "cifar10_cfm_euler50 (BASELINE configs[1] net, synthetic seeded weights)", "n_samples": int(x16.shape[0]), "nfe": a.nfe,
        "field": {"out2_gain": gain, "rms_v_t0_plain_weights": v_rms_plain, "rms_v_t0": v_rms,
                  "relative_move_|x1-x0|/|x0|_fp32": moves[("fp32", 0, a.nfe)]},
        "per_sample_rms_same_x0": {"bf16_vs_fp32": rms(x16 - x32), f"{a.nfe}_vs_{a.nfe - 1}_steps": rms(x32 - x32m),
                                   f"{a.nfe}_vs_{a.nfe // 2}_steps": rms(x32 - x32h)},
        "per_sample_bf16_vs_fp32": {
            "max_abs": float(err.max()), "rms": float((x16 - x32).pow(2).mean().sqrt()), "state_abs_max": float(x32.abs().max()),
            "worst_sample_rms": float(per_sample_rms.max()), "median_sample_rms": float(per_sample_rms.median()),
            "uint8_mean_abs_code_diff": float(code.float().mean()), "uint8_max_code_diff": int(code.max()),
            "uint8_fraction_equal": float((code == 0).float().mean()),
        },
        "frechet_proxy": {
            "feature_space": "evaluation.random_conv_features(seed=0): 3 x (conv3x3 + ReLU [+ avgpool2]), global mean+max -> 256-d",
            "bf16_vs_fp32_same_x0": evaluation.frechet_distance(f16, f32),
            "floor_fp32_vs_fp32_disjoint_x0": evaluation.frechet_distance(f32, f32b),
            "bf16_vs_fp32_disjoint_x0": evaluation.frechet_distance(f16, f32b),
            "scale_fp32_50step_vs_25step_same_x0": evaluation.frechet_distance(f32, f32h),
            "scale_fp32_50step_vs_49step_same_x0": evaluation.frechet_distance(f32, f32m),
        },
    }
    fp = res["frechet_proxy"]
    fp["fid_proxy_delta"] = fp["bf16_vs_fp32_disjoint_x0"] - fp["floor_fp32_vs_fp32_disjoint_x0"]
    fp["bf16_moves_the_distribution_less_than_one_euler_step"] = bool(fp["bf16_vs_fp32_same_x0"] < fp["scale_fp32_50step_vs_49step_same_x0"])
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


def split(a, net, sd, dev, nb, gain, v_rms):
    """Which roundings carry the bf16 mode's per-sample error (module docstring, round 4)."""
    def rounded(sel):
        out = dict(sd)
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() >= 3 and sel(k, v):   # conv [Co, Ci, k, k] and qkv / proj_out [Co, Ci, 1]; linears stay fp32 in both modes
                out[k] = v.to(torch.bfloat16).to(torch.float32)
        return out

    is3 = lambda k, v: v.dim() == 4 and v.shape[-1] == 3
    ends = lambda k, v: k in ("input_blocks.0.0.weight", "out.2.weight")
    sds = {"orig": sd, "w_all": rounded(lambda k, v: True), "w_3x3": rounded(is3), "w_1x1": rounded(lambda k, v: not is3(k, v)),
           "w_first_last": rounded(ends)}
    ts = torch.linspace(0, 1, a.nfe + 1).tolist()

    def run(precision, which, nfe=None, round_input=False):
        net.load_state_dict(sds[which])
        net.set_precision(precision)
        eng = net.engine(dev)
        tt = ts if nfe is None else torch.linspace(0, 1, nfe + 1).tolist()
        xs = []
        t0 = time.perf_counter()
        for k in range(nb):
            g = torch.Generator(device=dev).manual_seed(k)
            x = torch.randn(a.batch, 3, 32, 32, device=dev, generator=g)
            if not round_input:
                eng.cfm_euler(x, tt)
            else:   # the same Euler loop on the host: v(t_k, bf16(x_k)); x_{k+1} = x_k + (t_{k+1} - t_k) v in fp32
                for i in range(len(tt) - 1):
                    v = eng.forward(x.to(torch.bfloat16).to(torch.float32), float(tt[i]))
                    x = x + (tt[i + 1] - tt[i]) * v
            xs.append(x.cpu())
        torch.cuda.synchronize()
        print(f"{precision} {which} round_input={round_input} nfe={len(tt) - 1}: {nb * a.batch} samples in {time.perf_counter() - t0:.1f} s", flush=True)
        return torch.cat(xs)

    rms = lambda d: float(d.pow(2).mean().sqrt())
    ref = run("fp32", "orig")
    ref_r = run("fp32", "w_all")
    rows = {
        "bf16_mode_vs_fp32_mode (total)": rms(run("bf16", "orig") - ref),
        "weights_all_rounded (fp32 mode)": rms(ref_r - ref),
        "weights_3x3_rounded (fp32 mode)": rms(run("fp32", "w_3x3") - ref),
        "weights_1x1_qkv_proj_rounded (fp32 mode)": rms(run("fp32", "w_1x1") - ref),
        "weights_first_and_last_conv_rounded (fp32 mode)": rms(run("fp32", "w_first_last") - ref),
        "everything_but_weights: bf16 mode vs fp32 mode, both on bf16-representable weights": rms(run("bf16", "w_all") - ref_r),
        "network_input_rounded_each_evaluation (fp32 mode, host Euler loop)": rms(run("fp32", "orig", round_input=True) - ref),
        f"scale: {a.nfe}_vs_{a.nfe - 1}_steps (fp32 mode)": rms(run("fp32", "orig", nfe=a.nfe - 1) - ref),
        f"scale: {a.nfe}_vs_{a.nfe - 3}_steps (fp32 mode)": rms(run("fp32", "orig", nfe=a.nfe - 3) - ref),
        f"scale: {a.nfe}_vs_{a.nfe // 2}_steps (fp32 mode)": rms(run("fp32", "orig", nfe=a.nfe // 2) - ref),
    }
    tot, w, act = rows["bf16_mode_vs_fp32_mode (total)"], rows["weights_all_rounded (fp32 mode)"], rows["everything_but_weights: bf16 mode vs fp32 mode, both on bf16-representable weights"]
    res = {"workload": "cifar10_cfm_euler50 (BASELINE configs[1] net, synthetic seeded weights)", "n_samples": nb * a.batch, "nfe": a.nfe,
           "field": {"out2_gain": gain, "rms_v_t0": v_rms},
           "per_sample_rms_of_the_final_state_same_x0": rows,
           "reading": {"weights_share_of_variance": (w / tot) ** 2, "everything_else_share_of_variance": (act / tot) ** 2,
                       "sum_in_quadrature_over_total": ((w * w + act * act) ** 0.5) / tot}}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
