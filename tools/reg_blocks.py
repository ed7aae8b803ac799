"""How the resident loop's iteration time depends on the number of resident workgroups, for a cloud of --points points: the
multi-GPU kernel (reg_loop_kernel<true>) with a world of ONE rank (its mailbox stage included) and 256 / 128 / 64 / 32 / 16
workgroups.

    python tools/reg_blocks.py [--points 16384]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    from warpsense_amd.dist import HipGnBackend
    tau, res = 1000, 50
    view = W.DeviceMap([513, 513, 513], [256, 256, 256], None, (0, 0, 0))
    tsdf = W.TSDFCuda(view, tau, 640, res)
    pts = S.os1_128_scan()
    tsdf.update_tsdf(torch.from_numpy(pts).cuda(), (0, 0, 0), (0, 0, 32768))
    cloud = S.transform_points_mm(pts, S.perturbation(100, 100, 0, 5.0))
    cloud = np.ascontiguousarray(cloud[np.linspace(0, len(cloud) - 1, args.points).astype(np.int64)])
    q = torch.from_numpy(cloud).cuda()
    eye = np.eye(4, dtype=np.float32)
    for blocks in (256, 128, 64, 32, 16):
        ctx = W.Context(-1)
        rc = W.RegistrationCuda(ctx=ctx)
        rc.prepare_registration(q)
        b = HipGnBackend.__new__(HipGnBackend)
        b.reg, b.tsdf, b.res, b.flags, b._L, b.peers, b._pending = rc, tsdf, res, 0, rc._L, None, False
        b.connect_local([b], 0, blocks)
        for _ in range(3):
            T, it = b.register_peers(0, len(cloud), eye, 200, 0.1, 0.03)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            b.register_peers(0, len(cloud), eye, 200, 0.1, 0.03)
        ms = 1000.0 * (time.perf_counter() - t0) / args.reps
        print(f"{args.points} points, {blocks:3d} workgroups: {it} iterations, {ms:.3f} ms per registration, {1000.0 * ms / it:.2f} us per iteration (wall clock incl. launch)")
        rc.close()
        ctx.close()


if __name__ == "__main__":
    main()
