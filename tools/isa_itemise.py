#!/usr/bin/env python3
"""Itemise a kernel's ISA by loop and instruction class (VERDICT r3 task 7: where does the VALU of a kernel with DMA-only loaders sit?).

    hipcc ... --cuda-device-only -S csrc/conv_igemm.hip -o k.s ; python tools/isa_itemise.py k.s <mangled-name-substring> [out.json]

A loop = a label that a later branch jumps back to; every instruction is attributed to the INNERMOST loop that contains it (or to
"straight-line" code).  Classes: mfma, valu (v_* except mfma / readlane-type moves are counted too), salu, lds (ds_*), vmem (buffer_ /
global_ / scratch_), wait (s_waitcnt / s_nop / s_sleep / s_barrier), branch.  Static counts: a loop's dynamic weight is its trip count,
which the caller knows (e.g. the consumers' row loop runs once per two kernel rows)."""
import json, re, sys

def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    if op in ("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio") or op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    src, key = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    start = next(i for i, l in enumerate(lines) if key in l and re.match(r"^[A-Za-z_][\w$.]*:", l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end") or ".amdhsa_kernel" in lines[i])
    body = lines[start + 1:end]
    ins, labels = [], {}
    for l in body:
        t = l.strip()
        if not t or t.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", t):
            if re.match(r"^\.LBB\d+_\d+:", t): labels[t[:-1]] = len(ins)
            continue
        if re.match(r"^\.LBB\d+_\d+:", t):
            labels[t.split(":")[0]] = len(ins); continue
        op = t.split()[0]
        tgt = t.split()[-1] if op.startswith(("s_cbranch", "s_branch")) else None
        ins.append((op, tgt))
    loops = []   # (first, last) instruction index of every back edge's span
    for i, (op, tgt) in enumerate(ins):
        if tgt in labels and labels[tgt] <= i: loops.append((labels[tgt], i))
    loops.sort(key=lambda ab: ab[1] - ab[0])
    owner = [None] * len(ins)
    for k, (a, b) in enumerate(loops):
        for i in range(a, b + 1):
            if owner[i] is None: owner[i] = k
    out = {}
    for i, (op, _) in enumerate(ins):
        name = "straight-line" if owner[i] is None else f"loop@{loops[owner[i]][0]}..{loops[owner[i]][1]}"
        d = out.setdefault(name, {})
        c = classify(op)
        d[c] = d.get(c, 0) + 1
        if c == "valu":
            top = d.setdefault("valu_ops", {})
            top[op] = top.get(op, 0) + 1
    rows = sorted(out.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if k != "valu_ops"))
    for name, d in rows:
        tot = sum(v for k, v in d.items() if k != "valu_ops")
        tops = sorted(d.get("valu_ops", {}).items(), key=lambda kv: -kv[1])[:8]
        print(f"{name:28s} total {tot:5d} | " + " ".join(f"{k} {d.get(k, 0)}" for k in ("mfma", "valu", "salu", "lds", "vmem", "wait", "branch")) + " | valu: " + ", ".join(f"{o} {n}" for o, n in tops))
    if len(sys.argv) > 3:
        json.dump({"kernel": key, "regions": dict(rows)}, open(sys.argv[3], "w"), indent=1)

if __name__ == "__main__":
    main()
