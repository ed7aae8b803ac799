"""BASELINE.json configs[4] scale check: a 2049^3 TSDF map @ 20 mm (8.6e9 voxels > 2^32, ~217 GB on the device).

Size-independent property: the same scan integrated into the 2049^3 map and into a 1025^3 map (same resolution,
same window centre) must give identical voxels wherever the two windows overlap — the ring-buffer index of a world
voxel differs between the two maps and exceeds 32 bits in the large one.  Also times update + registration there.

    python tools/large_map_check.py [--big 2048] [--small 1024] [--res 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(W, S, torch, map_size, res, tau, pts, pert):
    size = (map_size,) * 3
    lm = W.DeviceMap([s + 1 - s % 2 for s in size], [s // 2 for s in size], None, (0, 0, 0))
    t0 = time.perf_counter()
    tsdf = W.TSDFCuda(lm, tau, 640, res)
    W.pause()
    t_alloc = time.perf_counter() - t0
    reg = W.RegistrationCuda(None)
    d_pts = torch.from_numpy(pts).cuda()
    d_pert = torch.from_numpy(pert).cuda()
    times = []
    for _ in range(3):
        W.pause()
        t1 = time.perf_counter()
        tsdf.update_tsdf(d_pts, (0, 0, 0), (0, 0, 32768))
        W.pause()
        t2 = time.perf_counter()
        reg.prepare_registration(d_pert)
        T, it = reg.register_cloud(tsdf.device_map(), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, res)
        t3 = time.perf_counter()
        times.append((t2 - t1, t3 - t2, it))
    return tsdf, T, times, t_alloc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=2048)
    ap.add_argument("--small", type=int, default=1024)
    ap.add_argument("--res", type=int, default=20)
    ap.add_argument("--room", type=float, nargs=3, default=(10.0, 8.0, 2.5), help="half extents (m); must fit the SMALL window")
    args = ap.parse_args()
    import torch
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    tau, res = 1000, args.res
    pts = S.os1_128_scan(half_extents_mm=tuple(1000.0 * r for r in args.room))  # default room: 1000 x 800 x 250 voxels at 20 mm
    assert max(args.room) * 1000.0 / res + 2 < (args.small + 1 - args.small % 2) // 2, "the room must fit the small window"
    pert = S.transform_points_mm(pts, S.perturbation(40, -30, 10, 1.5))
    half_small = (args.small + 1 - args.small % 2) // 2
    # compared region: the small window minus its outermost 2 voxel layers (a ray step whose on-ray voxel is outside
    # the window is skipped together with its in-window fan voxels, update_tsdf.cu:78, so the border may differ)
    lo = np.array([-(half_small - 2), -(half_small - 2), -200], dtype=np.int32)
    hi = np.array([half_small - 2, half_small - 2, 200], dtype=np.int32)

    out = {"res_mm": res}
    boxes = {}
    for name, msize in (("small", args.small), ("big", args.big)):
        tsdf, T, times, t_alloc = run(W, S, torch, msize, res, tau, pts, pert)
        box = tsdf.avg_map().extract_box(lo, hi)
        boxes[name] = box
        n = msize + 1 - msize % 2
        out[name] = {"map": n, "voxels": n ** 3, "alloc_and_fill_s": t_alloc, "update_ms": [1000 * t[0] for t in times],
                     "registration_ms": [1000 * t[1] for t in times], "iterations": [t[2] for t in times],
                     "touched_in_box": int(np.count_nonzero(box != int(W.pack_entry(tau, 0)))), "pose_t_mm": [float(v) for v in T[:3, 3]]}
        tsdf.close() if hasattr(tsdf, "close") else None
        del tsdf
        torch.cuda.empty_cache()
    out["overlap_voxels_compared"] = int(boxes["small"].size)
    out["identical"] = bool(np.array_equal(boxes["small"], boxes["big"]))
    print(json.dumps(out))
    if not out["identical"]:
        raise SystemExit("maps differ")


if __name__ == "__main__":
    main()
