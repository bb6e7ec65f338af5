#!/usr/bin/env python3
"""Idle time BETWEEN consecutive kernels of a rocprofv3 --kernel-trace run (rocpd SQLite database): for every dispatch, the gap from the previous
dispatch's end to its own start (same queue order = start order), summarised for the steady part of the run (the last `tail` fraction of the
dispatches) per following-kernel name and in total.  Tells how much of an evaluation is launch / drain / ramp between dependent kernels rather than
kernel time - the number a persistent multi-op kernel would have to beat.

    rocprofv3 --kernel-trace -d gpurun_out/gaps -o g -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/rocpd_gaps.py gpurun_out/gaps/g_results.db profiles/r5_kernel_gaps_bench_cifar_b256.json
"""
import collections
import json
import sqlite3
import sys


def main():
    dbp, outp = sys.argv[1:3]
    tail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    cur = sqlite3.connect(dbp).cursor()
    rows = sorted(cur.execute("select name, start, end from kernels").fetchall(), key=lambda r: r[1])
    rows = rows[int(len(rows) * (1.0 - tail)):]
    per = collections.defaultdict(list)
    busy = gap_total = 0
    for (pn, ps, pe), (n, s, e) in zip(rows[:-1], rows[1:]):
        g = s - pe
        per[n].append(g)
        gap_total += max(g, 0)
        busy += e - s
    span = rows[-1][2] - rows[0][2]
    allg = sorted(g for v in per.values() for g in v)
    q = lambda p: allg[min(len(allg) - 1, int(p * len(allg)))]
    out = {"dispatches": len(rows), "span_ms": span / 1e6, "kernel_ms": busy / 1e6, "gap_ms": gap_total / 1e6, "gap_fraction_of_span": gap_total / span,
           "gap_ns": {"p10": q(0.1), "median": q(0.5), "p90": q(0.9), "max": allg[-1], "negative (overlap)": sum(1 for g in allg if g < 0)},
           "by_following_kernel": {n[:110]: {"n": len(v), "mean_gap_ns": round(sum(v) / len(v), 1), "median_gap_ns": sorted(v)[len(v) // 2]}
                                   for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))}}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("dispatches", "span_ms", "kernel_ms", "gap_ms", "gap_fraction_of_span", "gap_ns")}))


if __name__ == "__main__":
    main()
