"""Replay a synthetic OS1-128 stream through warpsense_amd.App (BASELINE.json configs[2]: 10 Hz stream, 1024^3
sliding TSDF map @ 5 cm, one MI355X): sensor clouds in float metres -> device pre-processing -> TSDF update
(when the sensor moved > 0.3 m) -> Point-to-TSDF registration -> pose -> map shift (device-side slabs).

    python tools/replay_stream.py --map 1024 --scans 30 [--h5 /tmp/stream.h5]

Prints one JSON line: scans/s over the stream and the mean per-stage times (the reference's RuntimeEvaluator
forms "preprocess", "tsdf", "registration", "total")."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", type=int, default=1024)
    ap.add_argument("--res", type=int, default=50)
    ap.add_argument("--scans", type=int, default=30)
    ap.add_argument("--step", type=float, default=0.25, help="sensor motion per scan in metres (along x, half of it along y)")
    ap.add_argument("--shift", type=float, default=5.0, help="map shift distance in metres (map/shift)")
    ap.add_argument("--room", type=float, nargs=3, default=(22.0, 16.0, 2.5), help="half extents of the room in metres")
    ap.add_argument("--h5", default=None)
    ap.add_argument("--hz", type=float, default=0.0, help="pace the stream: scan k is handed over no earlier than k/hz seconds after the first "
                    "(0 = back to back; the sensor of configs[2] delivers 10 Hz, and the slab filing of an asynchronous shift "
                    "has the time between two shifts of a paced stream to finish)")
    ap.add_argument("--async-shift", action="store_true", help="map shift off the scan path (TSDFMapping.shift_map_async)")
    args = ap.parse_args()
    import warpsense_amd as W
    from warpsense_amd import synthetic as S

    size_m = args.map * args.res / 1000.0
    params = W.Params(W.MapParams(resolution=args.res, max_distance=1.0, max_weight=10, size=(size_m, size_m, size_m), shift=args.shift),
                      W.RegistrationParams(200, 0.1, 0.03))
    t0 = time.perf_counter()
    app = W.App(params, args.h5, async_shift=args.async_shift)
    t_setup = time.perf_counter() - t0
    he = tuple(1000.0 * r for r in args.room)
    clouds = []
    for k in range(args.scans):
        sensor = np.array([1000.0 * args.step * k, 500.0 * args.step * k, 0.0])
        pts = S.os1_128_scan(sensor_mm=tuple(sensor), half_extents_mm=he, seed=1000 + k)
        clouds.append(((pts.astype(np.float64) - sensor) / 1000.0).astype(np.float32))
    W.pause()
    t1 = time.perf_counter()
    busy = 0.0
    for k, c in enumerate(clouds):
        if args.hz > 0.0:
            wait = t1 + k / args.hz - time.perf_counter()
            if wait > 0.0:
                time.sleep(wait)
        tb = time.perf_counter()
        app.cloud_callback(c)
        busy += time.perf_counter() - tb
    W.pause()
    t2 = time.perf_counter()
    stages = {}
    for key in ("preprocess", "tsdf", "registration", "total"):
        vals = [t[key] for t in app.timings if key in t]
        stages[key + "_ms"] = 1000.0 * float(np.mean(vals)) if vals else None
    true_last = np.array([1000.0 * args.step * (args.scans - 1), 500.0 * args.step * (args.scans - 1), 0.0])
    t3 = time.perf_counter()
    app.terminate()
    t4 = time.perf_counter()
    print(json.dumps({"workload": f"{args.scans} synthetic OS1-128 scans (131072 pts), {args.map}^3 sliding map @ {args.res} mm, App replay",
                      "args": {"step_m": args.step, "shift_m": args.shift, "room_m": list(args.room), "h5": bool(args.h5), "hz": args.hz},
                      "scans_per_s": args.scans / (t2 - t1), "stream_s": t2 - t1, "callback_busy_s": busy, "setup_s": t_setup, **stages,
                      "tsdf_updates": app.n_updates, "map_shifts": app.n_shifts, "async_shift": bool(args.async_shift),
                      "slowest_scan_ms": 1000.0 * float(max(t["total"] for t in app.timings[2:])),
                      "scans_over_100ms": int(sum(1 for t in app.timings[2:] if t["total"] > 0.1)),
                      "points_after_preprocess": float(np.mean([t["points"] for t in app.timings])),
                      "iterations_mean": float(np.mean([t["iterations"] for t in app.timings])),
                      "final_position_error_mm": float(np.linalg.norm(app.poses[-1][:3, 3] - true_last)),
                      "terminate_write_back_s": t4 - t3, "h5": args.h5}))


if __name__ == "__main__":
    main()
