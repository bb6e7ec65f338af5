#!/usr/bin/env python3
"""Pretty-print a bench.py --profile-out JSON (per-op HIP-event timings of one forward)."""
import json, sys
d = json.load(open(sys.argv[1]))
agg = {}
for r in d["ops"]:
    if r["kind"] == "conv":
        key = f"conv k{r['ks']} {r['cin']:4d}->{r['cout']:4d} {r['h']:3d}x{r['w']:<3d} tile{tuple(r['tile'])}"
    else:
        key = f"{r['kind']} c{r['cin']} {r['h']}x{r['w']}"
    a = agg.setdefault(key, dict(n=0, ms=0.0, flops=0.0, bytes=0.0))
    a["n"] += 1; a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]
tot = sum(a["ms"] for a in agg.values())
print(f"forward {tot:.3f} ms, batch {d['batch']} {d['precision']}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    tf = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0
    gb = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0
    print(f"{k:52s} x{a['n']:2d} {a['ms']*1e3/a['n']:8.1f} us each {a['ms']:7.3f} ms ({100*a['ms']/tot:4.1f}%) {tf:7.1f} TF/s {gb:7.1f} GB/s")
