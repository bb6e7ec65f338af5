#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (one pass = one rocpd database).

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY ... -d gpurun_out/pmc_sq -o s -- python3 bench.py --steps 1 --warmup 0 --nfe 2 --no-cpu-baseline
    python tools/pmc_kernels.py gpurun_out/pmc_sq/s_results.db [out.json]

Prints, per kernel (short name), the number of dispatches and the per-dispatch average of every counter found.
"""
import collections
import json
import re
import sqlite3
import sys

NAMES = (r"(attn_fused_pers_kernel|attn_fused_kernel|conv3x3_ws_kernel|conv3x3_pp_kernel|conv1x1_pp_kernel|conv3x3_small_kernel|conv3x3_out_kernel|conv3x3_in_kernel|conv_igemm_kernel|conv1x1_kernel|gn_affine_kernel|gn_finalize_kernel|attention_kernel|"
         r"affine_pool_kernel|linear_small_kernel|linear_kernel|timestep_embedding_kernel|pack_nhwc_kernel|unpack_nchw_kernel|"
         r"resample\w*_kernel|ew4\w*|rk_\w+_kernel)")


def short(name):
    m = re.search(NAMES, name)
    if not m:
        return re.sub(r"\(.*", "", name)[:60]
    base = m.group(1)
    t = re.search(base + r"<([^>]*)>", name) or re.search(base + r"I([A-Za-z0-9_]*)E", name)
    return base + ("<" + t.group(1)[:40] + ">" if t else "")


def main():
    dbp = sys.argv[1]
    cur = sqlite3.connect(dbp).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tabs:
        raise SystemExit(f"{dbp}: no counters_collection view; tables: {tabs}")
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for name, cname, val in cur.execute(f"select {kcol}, counter_name, value from counters_collection"):
        a = agg[short(name)][cname]
        a[0] += float(val); a[1] += 1
    res = {}
    for k, cs in agg.items():
        res[k] = {"dispatches": max(v[1] for v in cs.values())}
        res[k].update({c: v[0] / v[1] for c, v in cs.items()})
    order = sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", res[k].get("GRBM_GUI_ACTIVE", 0)) * res[k]["dispatches"])
    for k in order[:14]:
        print(k, {c: (round(v) if isinstance(v, float) else v) for c, v in res[k].items()})
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
