#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration, share) of a rocprofv3 --kernel-trace --stats run.
This rocprofv3 writes a rocpd SQLite database (<dir>/<name>_results.db); the summary is what `--stats` would print.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db profiles/r1_rocprofv3_kernel_stats_bench_cifar_b256.csv
"""
import collections
import csv
import sqlite3
import sys


def main():
    dbp, outp = sys.argv[1:3]
    cur = sqlite3.connect(dbp).cursor()
    agg = collections.defaultdict(list)
    for name, s, e in cur.execute("select name, start, end from kernels"):
        agg[name].append(e - s)
    total = sum(sum(v) for v in agg.values())
    with open(outp, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3), min(v), max(v)])
    print(f"{len(agg)} kernels, {sum(len(v) for v in agg.values())} dispatches, {total / 1e6:.1f} ms of GPU time -> {outp}")


if __name__ == "__main__":
    main()
