#!/usr/bin/env python3
"""The multi-GPU resident loop (device-side exchange of the 44 sums through mailboxes, ws_register_cloud_peers) exercised on
ONE GPU: R ranks as ws_reg handles on their own contexts / streams of this process, 256 // R resident workgroups each, every
rank registering its shard of the benchmark cloud against the 513^3 map.  Prints one JSON line.

    python tools/peer_bench.py [--ranks 2] [--reps 10]

Own process on purpose: a process drives a handful of hardware queues, and the ranks' kernels must be on the chip together
(bench.py, which has torch / RCCL streams of its own, runs this as a subprocess)."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# One process maps its streams onto 4 hardware queues by default; ranks whose streams share a queue cannot be on the chip
# together (measured: 3 ranks time out with the default, 8 ranks work with 16 queues).  One process per GPU -- the deployment --
# has one such loop per process and does not need this.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--map", type=int, default=512)
    args = ap.parse_args()
    import torch
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    from warpsense_amd.dist import HipGnBackend, shard_range
    tau, mw, res = 1000, 640, 50
    size = (args.map,) * 3
    reg_params = (200, 0.1, 0.03)
    ctx = W.Context(0)
    lm = W.LocalMap(*size, tau, 0)
    host_map = lm.device_map()
    host_map.data_ = None
    tsdf = W.TSDFCuda(host_map, tau, mw, res, ctx)
    points = S.os1_128_scan()
    perturbed = S.transform_points_mm(points, S.perturbation())
    d_points = torch.from_numpy(points).cuda()
    d_pert = torch.from_numpy(perturbed).cuda()
    tsdf.update_tsdf(d_points, (0, 0, 0), (0, 0, 32768))
    ctx.sync()
    n = points.shape[0]
    eye = np.eye(4, dtype=np.float32)
    one = W.RegistrationCuda(None, ctx)
    one.prepare_registration(d_pert)
    T1, it1 = one.register_cloud(tsdf.device_map(), eye, *reg_params, res)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        one.register_cloud(tsdf.device_map(), eye, *reg_params, res)
    one_ms = 1000.0 * (time.perf_counter() - t0) / args.reps
    R = args.ranks
    blocks = (256 // R) // 8 * 8
    ranks = []
    for r in range(R):
        c = W.Context(0)
        rc = W.RegistrationCuda(None, c)
        rc.prepare_registration(d_pert)
        c.sync()
        b = HipGnBackend.__new__(HipGnBackend)
        b.reg, b.tsdf, b.res, b.flags, b._L, b.peers, b._pending = rc, tsdf, res, 0, rc._L, None, False
        ranks.append(b)
    for r, b in enumerate(ranks):
        b.connect_local(ranks, r, blocks)
    out = [None] * R

    def run_rank(r, reps):
        first, count = shard_range(n, r, R)
        for _ in range(reps):
            out[r] = ranks[r].register_peers(first, count, eye, *reg_params)

    def all_ranks(reps):
        th = [threading.Thread(target=run_rank, args=(r, reps)) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    all_ranks(2)
    t0 = time.perf_counter()
    all_ranks(args.reps)
    dt = time.perf_counter() - t0
    if any(o is None for o in out):
        print(json.dumps({"error": "exchange timed out", "ranks": R, "blocks_per_rank": blocks}))
        return
    same = all(o[1] == it1 and np.array_equal(o[0], T1) for o in out)
    print(json.dumps({"ranks": R, "blocks_per_rank": blocks, "ms_per_registration": 1000.0 * dt / args.reps, "iterations": out[0][1],
                      "us_per_iteration": 1e6 * dt / args.reps / max(out[0][1], 1), "same_result_as_one_rank": bool(same),
                      "one_rank_ms_per_registration": one_ms, "one_rank_us_per_iteration": 1000.0 * one_ms / max(it1, 1),
                      "note": f"{R} ranks x {blocks} resident workgroups on ONE GPU, wall clock per registration incl. launch + host wake-up"}))


if __name__ == "__main__":
    main()
