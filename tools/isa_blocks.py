#!/usr/bin/env python3
"""Basic blocks of one kernel's ISA (hipcc -S output cut to one function): label, instruction counts by class, where it
branches.  Backward branches (loops) are marked.    python tools/isa_blocks.py free.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
blocks, cur = [], None
order = {}
for ln in lines:
    s = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+|_ZN\S+):", s)
    if m:
        cur = {"label": m.group(1), "v": 0, "s": 0, "lds": 0, "vmem": 0, "other": 0, "br": [], "rl": 0, "scr": 0}
        order[cur["label"]] = len(blocks)
        blocks.append(cur)
        continue
    if cur is None or not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        cur["rl"] += 1
    if op.startswith("scratch_"):
        cur["scr"] += 1
        cur["vmem"] += 1
    elif op.startswith("v_"):
        cur["v"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur["br"].append(s.split()[-1])
        cur["s"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        cur["vmem"] += 1
    else:
        cur["other"] += 1
for i, b in enumerate(blocks):
    back = [t for t in b["br"] if t in order and order[t] <= i]
    print(f"{i:4d} {b['label']:<14} v={b['v']:<4} s={b['s']:<4} lds={b['lds']:<3} vmem={b['vmem']:<3} lane={b['rl']:<3} scr={b['scr']:<2} -> {','.join(b['br'])}"
          + (f"   <== LOOP to {','.join(back)}" if back else ""))
