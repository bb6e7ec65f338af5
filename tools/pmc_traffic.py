#!/usr/bin/env python3
"""HBM traffic per launch of each kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db profiles/r1_pmc_hbm_traffic.json

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3 reports both counters in kilobytes (x 1024 B);
on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) coalesced reads at 64 B, so fetched bytes =
FETCH_SIZE x 1024 x 2; WRITE_SIZE x 1024 is exact for 16-B-per-lane stores.  Infinity-Cache hits are counted, not excluded.
"""
import collections
import json
import re
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    out = collections.defaultdict(lambda: [0.0, 0])
    if view:
        cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        for name, cname, val in cur.execute(f"select {kcol}, counter_name, value from {view}"):
            if cname != counter:
                continue
            k = short(name)
            out[k][0] += float(val); out[k][1] += 1
    else:
        raise SystemExit(f"{db_path}: no counters_collection view; tables: {tabs}")
    return out


def short(name):
    """base kernel name of a mangled or demangled symbol (instantiations of one template are summed)"""
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_]\w*?_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"([A-Za-z_]\w*_kernel|ew4\w*|__amd_rocclr_\w+)", re.sub(r"^void\s+", "", name))
    return m.group(1) if m else name[:60]


def main():
    fdb, wdb, outp = sys.argv[1:4]
    f = per_kernel(fdb, "FETCH_SIZE")
    w = per_kernel(wdb, "WRITE_SIZE")
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 0",
           "correction": "bytes = FETCH_SIZE*1024*2 (gfx950: wide reads tallied at half) + WRITE_SIZE*1024", "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fs, fn = f.get(k, [0.0, 0]); ws, wn = w.get(k, [0.0, 0])
        n = max(fn, wn)
        if n == 0:
            continue
        fetch = fs * 1024.0 * 2.0 / max(fn, 1); write = ws * 1024.0 / max(wn, 1)
        res["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    json.dump(res, open(outp, "w"), indent=1)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k:28s} x{v['launches']:6d}  fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
