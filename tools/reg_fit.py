"""Fixed and per-iteration cost of the resident registration loop: reg_loop_kernel timed (hipEvents on the library's
stream) for several max_iterations on the benchmark scan, least-squares line through the points.

    python tools/reg_fit.py [--points N]      (N: only the first N points of every ring-interleaved selection of the cloud)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=0, help="register a subset of this many points (spread evenly over the scan)")
    args = ap.parse_args()
    import torch
    import warpsense_amd as W
    from warpsense_amd import _lib
    from warpsense_amd import synthetic as S
    tau, res = 1000, 50
    view = W.DeviceMap([513, 513, 513], [256, 256, 256], None, (0, 0, 0))
    tsdf = W.TSDFCuda(view, tau, 640, res)
    pts = S.os1_128_scan()
    tsdf.update_tsdf(torch.from_numpy(pts).cuda(), (0, 0, 0), (0, 0, 32768))
    reg = W.RegistrationCuda(None)
    cloud = S.transform_points_mm(pts, S.perturbation(100, 100, 0, 5.0))
    if args.points:
        cloud = np.ascontiguousarray(cloud[np.linspace(0, len(cloud) - 1, args.points).astype(np.int64)])
    print(f"cloud: {len(cloud)} points")
    q = torch.from_numpy(cloud).cuda()
    reg.prepare_registration(q)
    ctx = tsdf.ctx
    eye = np.eye(4, dtype=np.float32)
    xs, ys = [], []
    for max_it in (1, 2, 5, 10, 20, 50, 100, 150):
        for _ in range(3):
            reg.register_cloud(tsdf.device_map(), eye, max_it, 0.1, 0.03, res)
        ctx.prof_reset()
        ctx.prof_enable(1 << _lib.WS_K_REG)
        its = []
        for _ in range(10):
            _, it = reg.register_cloud(tsdf.device_map(), eye, max_it, 0.1, 0.03, res)
            its.append(it)
        torch.cuda.synchronize()
        ms, cnt = ctx.prof_read(_lib.WS_K_REG)
        ctx.prof_enable(0)
        us = 1000.0 * ms / cnt
        print(f"max_iterations {max_it:4d}: iterations {its[0]:4d}, reg_loop_kernel {us:8.1f} us")
        xs.append(its[0])
        ys.append(us)
    slope, intercept = np.polyfit(np.array(xs, dtype=np.float64), np.array(ys), 1)
    print(f"fit: {intercept:.1f} us fixed + {slope:.3f} us per iteration")


if __name__ == "__main__":
    main()
