"""Randomised parity soak of the TSDF update against the CPU oracle (bit-exact avg_map after 1-2 scans): random map
sizes, resolutions, truncation distances, sensor positions (incl. off-centre windows), rooms larger and smaller than the
window, tilted `up` vectors.  The fixed-seed cases live in tests/test_gpu_tsdf.py; this is for changes to the ray
arithmetic (ws_march.h, ray_setup_kernel).

    python tools/soak_tsdf.py [--cases 30] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        res = int(rng.choice([16, 20, 32, 50, 64, 100]))
        tau = int(rng.choice([res * 6, res * 10, res * 20]))
        size = tuple(int(rng.integers(20, 70)) * 2 for _ in range(3))
        mw = 640
        lm = W.LocalMap(size[0], size[1], size[2], tau, 0)
        oa = O.OracleMap(size, tau, 0)
        on = oa.copy()
        t = W.TSDFCuda(lm.device_map(), tau, mw, res)
        ext = np.array(size, dtype=np.float64) * res
        he = ext * rng.uniform(0.2, 0.7, 3)  # rooms smaller and larger than the window (half extent 0.5)
        sensor_vox = np.array([int(rng.integers(-s // 5, s // 5 + 1)) for s in size])
        up = (0, 0, 32768) if rng.random() < 0.6 else tuple(int(v) for v in np.round(32768 * np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 0.97])))
        ok = True
        for scan in range(int(rng.integers(1, 3))):
            sensor_mm = tuple((sensor_vox * res + res // 2).astype(np.float64) + rng.uniform(-5, 5, 3))
            sensor_mm = tuple(np.clip(sensor_mm, -he * 0.9, he * 0.9))
            pts = S.os1_128_scan(sensor_mm=sensor_mm, rings=int(rng.choice([8, 16, 32])), azimuths=int(rng.choice([64, 128, 256])),
                                 half_extents_mm=tuple(he), seed=int(rng.integers(1, 1 << 30)), yaw_rad=float(rng.uniform(0, 6.28)))
            sp = tuple(int(np.floor(v / res)) for v in sensor_mm)
            O.update_tsdf(oa, on, pts, sp, up, tau, mw, res)
            t.update_tsdf(torch.from_numpy(pts).cuda(), sp, up)
        host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
        t.avg_map().to_host(host)
        diff = int((host.data_ != oa.data).sum())
        if diff:
            ok = False
            bad += 1
        st = t.stats(raise_on_error=False)
        print(f"case {case:3d}: res {res:3d} tau {tau:5d} size {size} up {up} records {st['records']:8d} touched tiles {st['tiles']:6d} "
              f"errors {st['error_flags']} -> {'ok' if ok else 'DIFF ' + str(diff)}")
    print(f"{args.cases} cases, {bad} with differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
