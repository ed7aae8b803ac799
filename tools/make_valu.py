#!/usr/bin/env python3
"""profiles/pmc_valu.json: vector instructions per launch of the update's kernels (SQ_INSTS_VALU), from the summary a counter pass
of tools/ab_sq.sh leaves in gpurun_out/sq_<variant>.txt -- what bench.py turns into `lane_instructions.per_scatter_target`.

    python tools/make_valu.py gpurun_out/sq_a_head.txt
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {"march_tail_kernel": "march_tails", "march_free_kernel": "march_free", "tile_resolve_kernel<false, true>": "tile_resolve",
         "ray_setup_kernel": "ray_setup", "ray_sort_kernel": "ray_sort"}
out = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(.*?)\s+(SQ_\w+)\s+(\d+)\s+([\d.]+)\s", line)
    if not m or m.group(2) != "SQ_INSTS_VALU":
        continue
    for frag, key in names.items():
        if frag in m.group(1):
            out[key] = float(m.group(4))
sha = os.environ.get("WS_GIT_SHA")
if not sha:
    try:
        sha = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        sha = "unknown"
res = {"valu_per_launch": out, "git_sha": sha, "source": os.path.basename(sys.argv[1])}
with open(os.path.join(ROOT, "profiles", "pmc_valu.json"), "w") as fh:
    json.dump(res, fh, indent=1)
print(res)
