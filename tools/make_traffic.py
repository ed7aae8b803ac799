#!/usr/bin/env python3
"""profiles/pmc_traffic.json: HBM bytes per launch of the kernels bench.py reports a roofline for, from the
separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (values are KB, summed over the hardware instances of a
dispatch, averaged over the dispatches of a kernel).

Corrections (MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE counts a wide coalesced stream (16 B per lane) at half
its bytes (128-byte requests tallied at 64).  Round 6 calibrated the load shapes of the scatter's kernels (tools/read_calib.hip,
profiles/r06_read_calib.txt): coalesced 4, 8 and 16 bytes per lane, and 8- / 16-byte lanes in 256-byte runs at random places
(a sub-chunk per half-wave, a tile's z-runs: tile_resolve's reads) ALL count exactly 0.50 of the bytes loaded; a lone 8-byte load
counts 64 B, i.e. one request.  So every request is tallied at half a 128-byte line and FETCH_SIZE is DOUBLED for every kernel
(until round 5: for integrate_dense only, which left tile_resolve's reads -- 140 MB counted for at least 178 MB of records and
map entries -- unexplained; VERDICT r5 weak #10a).  WRITE_SIZE as calibrated in round 5 (tools/write_calib.hip): the bytes of a
dense store, 32 B per scattered store.

    python tools/make_traffic.py gpurun_out/prof_r02  ->  profiles/pmc_traffic.json
"""
import glob
import json
import os
import sqlite3
import sys


def per_kernel(pattern):
    dbs = sorted(glob.glob(pattern + "/*.db") + glob.glob(pattern + "/*/*.db"))
    if not dbs:
        return {}
    con = sqlite3.connect(dbs[0])
    q = ("select name, avg(v) from (select name, dispatch_id, sum(counter_value) as v from pmc_events group by name, dispatch_id) "
         "group by name")
    return {name: avg for name, avg in con.execute(q)}


def main(prefix):
    out = {}
    for mode in ("sparse", "dense"):
        f = per_kernel(f"{prefix}_pmc_FETCH_SIZE_{mode}")
        w = per_kernel(f"{prefix}_pmc_WRITE_SIZE_{mode}")
        if not f or not w:
            continue

        def kb(d, frag):
            return sum(v for k, v in d.items() if frag in k)
        # the resolve of THIS route only (the sparse pass also ran bench.py's dense-equivalent leg, whose
        # tile_resolve_kernel<false, false> was summed into the sparse figure -- 891.6 MB instead of 669.9; the passes now set
        # WS_BENCH_SKIP_DENSE_EQ=1 as well, so a pass holds the kernels of one route)
        resolve = "tile_resolve_kernel<false, true>" if mode == "sparse" else "tile_resolve_kernel<false, false>"
        for name, frag in (("march_tails", "march_tail_kernel"), ("march_free", "march_free_kernel"), ("tile_resolve", resolve),
                           ("ray_setup", "ray_s")):
            val = (2 * kb(f, frag) + kb(w, frag)) * 1024  # FETCH_SIZE: half-counted (see above)
            out[f"{name}:{mode}"] = int(val)
        scatter = sum(out[f"{k}:{mode}"] for k in ("march_tails", "march_free", "tile_resolve", "ray_setup"))
        out[f"scatter_total:{mode}"] = int(scatter)
        if mode == "dense":
            out["integrate:dense"] = int((2 * kb(f, "integrate_dense") + kb(w, "integrate_dense")) * 1024)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # which sources these passes ran on: the commit (GRAFT snapshots carry no .git: WS_GIT_SHA from the caller) and a hash of
    # the kernel sources themselves
    import hashlib
    import subprocess
    sha = os.environ.get("WS_GIT_SHA")
    if not sha:
        try:
            sha = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            sha = "unknown"
    h = hashlib.sha256()
    for f_ in ("tsdf_update.hip", "tsdf_integrate.hip", "ws_march.h", "ws_dda.h", "ws_device.h", "registration.hip", "api.hip"):
        with open(os.path.join(root, "warpsense_amd", "csrc", f_), "rb") as fh:
            h.update(fh.read())
    out["git_sha"] = sha
    out["csrc_sha256_16"] = h.hexdigest()[:16]
    with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(out)


if __name__ == "__main__":
    main(sys.argv[1])
