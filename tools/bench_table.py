#!/usr/bin/env python3
"""Markdown rows of DESIGN.md section 6's bench table from a bench_lines jsonl (one bench.py line per row)."""
import json, sys
print("| workload | mode | images/s | at the reference box | ms per network evaluation | launches | whole-path MFMA fraction | `conv3x3_ws_kernel` (dominant): rate = fraction (avg launch, share of the forward) | `conv3x3_pp_kernel` | 1x1 convs | attention block |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]; wp = r["whole_path"]; o = r["other_kernels"]
    name = d["config"]["workload"]
    b = d["config"].get("images_per_gpu_per_step")
    if name == "cifar10_cfm_euler50_b256" and b and b != 256: name += f" (B = {b})"
    pp, k1, at = o.get("conv3x3_pp_kernel"), o.get("conv1x1"), o.get("attention_block")
    f = lambda e, unit="PFLOP/s": "-" if not e else (f"{e['achieved_tflops']/1e3:.2f} PFLOP/s = {e['frac']:.2f} ({e['launches']} x {e['avg_launch_us']:.0f} us, {100*e['share_of_forward']:.0f} %)" if e["bound"] == "mfma" else f"{e['achieved_gbs']/1e3:.2f} TB/s ({e['launches']} x {e['avg_launch_us']:.1f} us, {100*e['share_of_forward']:.0f} %)")
    print(f"| {name} | {d['dtype']} | **{d['value']:.1f}** | {r['box']['value_at_reference_box']:.1f} | {wp['ms_per_network_evaluation']:.2f} | {wp['launches_per_evaluation']} | {wp['achieved_mfma_frac']:.2f} | {r['achieved']/1e3:.2f} PFLOP/s = {r['frac']:.2f} ({r['avg_launch_us']:.0f} us, {100*r['share_of_forward']:.0f} %) | {f(pp)} | {f(k1)} | {at['avg_launch_us']:.1f} us |" if at else f"| {name} | {d['dtype']} | **{d['value']:.1f}** | {r['box']['value_at_reference_box']:.1f} | {wp['ms_per_network_evaluation']:.2f} | {wp['launches_per_evaluation']} | {wp['achieved_mfma_frac']:.2f} | {r['achieved']/1e3:.2f} PFLOP/s = {r['frac']:.2f} ({r['avg_launch_us']:.0f} us, {100*r['share_of_forward']:.0f} %) | {f(pp)} | {f(k1)} | - |")
