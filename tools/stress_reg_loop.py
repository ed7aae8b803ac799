"""Stress the resident registration loop (grid barrier + atomic group sums): the same cloud registered N times must give
bit-identical poses and iteration counts every time, in the resident mode and against one-launch-per-iteration.

    python tools/stress_reg_loop.py [--n 2000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    args = ap.parse_args()
    import torch
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    tau, res = 1000, 50
    view = W.DeviceMap([513, 513, 513], [256, 256, 256], None, (0, 0, 0))
    tsdf = W.TSDFCuda(view, tau, 640, res)
    pts = S.os1_128_scan()
    tsdf.update_tsdf(torch.from_numpy(pts).cuda(), (0, 0, 0), (0, 0, 32768))
    reg = W.RegistrationCuda(None)
    clouds = [torch.from_numpy(S.transform_points_mm(pts, S.perturbation(100 - 30 * k, 100 + 10 * k, 5 * k, 5.0 - k))).cuda() for k in range(4)]
    ref = []
    for c in clouds:
        reg.set_loop(W.WS_REG_LOOP_LAUNCHES)
        reg.prepare_registration(c)
        ref.append(reg.register_cloud(tsdf.device_map(), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, res))
    reg.set_loop(W.WS_REG_LOOP_RESIDENT)
    bad = 0
    for i in range(args.n):
        k = i % len(clouds)
        reg.prepare_registration(clouds[k])
        T, it = reg.register_cloud(tsdf.device_map(), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, res)
        if it != ref[k][1] or not np.array_equal(T, ref[k][0]):
            bad += 1
            print("mismatch at", i, it, ref[k][1])
    print(f"{args.n} resident registrations, iterations {[r[1] for r in ref]}, mismatches: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
