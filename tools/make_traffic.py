#!/usr/bin/env python3
"""profiles/pmc_traffic.json: HBM bytes per launch of the kernels bench.py reports a roofline for, from the
separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (values are KB).

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts a wide coalesced stream at half its bytes on gfx950 —
doubled for integrate_dense (128-bit streaming loads, verified: 2 x 527 MB + 1055 MB written == 16 B/voxel).
For the scattered 8-byte accesses of the march kernels the counter is uncalibrated: recorded as is.

    python tools/make_traffic.py gpurun_out/prof_r01e
"""
import json
import os
import sqlite3
import sys


def per_kernel(db):
    con = sqlite3.connect(db)
    return {name: avg for name, avg in con.execute("select name, avg(counter_value) from pmc_events group by name")}


def main(prefix):
    out = {}
    for mode in ("sparse", "dense"):
        f = per_kernel(f"{prefix}_pmc_fetch_{mode}/pmc_results.db")
        w = per_kernel(f"{prefix}_pmc_write_{mode}/pmc_results.db")

        def kb(d, frag):
            return sum(v for k, v in d.items() if frag in k)
        march = (kb(f, "march_kernel<2") + kb(f, "march_kernel<3") + kb(w, "march_kernel<2") + kb(w, "march_kernel<3") +
                 kb(f, "ray_") + kb(w, "ray_")) * 1024
        out[f"march_emit:{mode}:global"] = int(march)
        if mode == "dense":
            out["integrate:dense:global"] = int((2 * kb(f, "integrate_dense") + kb(w, "integrate_dense")) * 1024)
        else:
            out["integrate:sparse:global"] = int((kb(f, "integrate_sparse") + kb(w, "integrate_sparse")) * 1024)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(out)


if __name__ == "__main__":
    main(sys.argv[1])
