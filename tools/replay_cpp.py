"""BASELINE configs[2] with the C++ node: writes the synthetic OS1-128 stream of tools/replay_stream.py (same rooms, same motion) as
float32 sensor clouds and runs examples/replay_bench (warpsense::App) on it -- paced at the sensor's rate and back to back, with
the map shift off the scan path.  One JSON line per run (wall clock per stage + the device's own clock for the same scans).

    python tools/replay_cpp.py --map 1024 --scans 60 --out profiles/r06_replay_1024_cpp
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", type=int, default=1024)
    ap.add_argument("--res", type=int, default=50)
    ap.add_argument("--scans", type=int, default=60)
    ap.add_argument("--step", type=float, default=0.25)
    ap.add_argument("--shift", type=float, default=2.0)
    ap.add_argument("--room", type=float, nargs=3, default=(20.0, 16.0, 5.0))
    ap.add_argument("--hz", type=float, nargs="+", default=[10.0, 0.0])
    ap.add_argument("--sync-shift", action="store_true")
    ap.add_argument("--out", default=None, help="prefix of the JSON files (<prefix>_paced.json / <prefix>_b2b.json)")
    args = ap.parse_args()
    from warpsense_amd import synthetic as S
    exe = os.path.join(ROOT, "examples", "replay_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    he = tuple(1000.0 * r for r in args.room)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "clouds.bin")
        n = None
        with open(path, "wb") as f:
            for k in range(args.scans):
                sensor = np.array([1000.0 * args.step * k, 500.0 * args.step * k, 0.0])
                pts = S.os1_128_scan(sensor_mm=tuple(sensor), half_extents_mm=he, seed=1000 + k)
                n = pts.shape[0]
                f.write(((pts.astype(np.float64) - sensor) / 1000.0).astype(np.float32).tobytes())
        env = dict(os.environ)
        if not args.sync_shift:
            env["WS_REPLAY_ASYNC_SHIFT"] = "1"
        for hz in args.hz:
            r = subprocess.run([exe, path, str(args.scans), str(n), str(args.map), str(args.res), "1000", "640", str(args.shift), str(hz)],
                               capture_output=True, text=True, env=env, timeout=1800)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                print(r.stdout[-2000:], r.stderr[-2000:], file=sys.stderr)
                raise SystemExit(1)
            d = json.loads(lines[-1])
            d["args"] = {"step_m": args.step, "shift_m": args.shift, "room_m": list(args.room)}
            print(json.dumps(d))
            if args.out:
                with open(f"{args.out}_{'paced' if hz > 0 else 'b2b'}.json", "w") as f:
                    f.write(json.dumps(d) + "\n")


if __name__ == "__main__":
    main()
