#!/usr/bin/env python3
"""USEFUL lanes per issued wave instruction of the two march kernels, counted from the benchmark scan itself (numpy, no GPU).

The SQ counters (tools/pmc_lanes.sh) say how many lanes the EXEC mask leaves on; the branch-free phases of the marches keep
lanes on whose results are thrown away, so this model walks the synthetic OS1-128 scan with the work decomposition of the
kernels (warpsense_amd/csrc/tsdf_update.hip) and counts, per phase, lane-slots that carry a live unit of work:

  free pass   64 rays x 4 lanes per workgroup, lane c walks the steps [c*ch, (c+1)*ch) of the free-space part
              sample phase: live = the lane still has samples; emit phase: batches of 64 queued candidates
  tail march  64 sorted rays x 8 parts (2 workgroups x 4 waves), wave w walks part w of all 64 rays
              sample phase as above; emit phase: batches of 64 queued samples, then ROUNDS of at most one scatter target per lane
              (round -1: the on-ray target of every sample with a non-zero weight; round j: fan step j of the samples whose fan
              has more than j steps, j != mid)

and the same for the alternatives (--alt): a free pass that iterates over column changes instead of samples, tail rounds
with the fan steps spread over the lanes.

    python tools/lane_model.py [--alt]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warpsense_amd import synthetic as S  # noqa: E402

RES, TAU, MAP = 50, 1000, 513
HALF = RES // 2
DZ = 100


def tdiv(a, b):
    """C division (truncating) of int64 arrays."""
    q = np.abs(a) // np.abs(b)
    return np.where((a < 0) != (b < 0), -q, q)


def ray_setup(points):
    pos = np.array([HALF, HALF, HALF], dtype=np.int64)
    d = points.astype(np.int64) - pos
    dist = np.sqrt((d * d).sum(1).astype(np.float32)).astype(np.int64)
    steps = (dist + TAU - 1) // HALF + 1
    len_neg = ((RES + 1) // 2 * 32768 + DZ - 1) // DZ
    slack = 2 * ((RES + 1) // 2 + 1) + 3 * RES + 4
    keyed_len = np.minimum(len_neg, dist - TAU - slack)
    kfirst = np.where(keyed_len > 1, np.maximum(0, (keyed_len - 1) // HALF - 1), 0)
    kfirst = np.minimum(kfirst, steps)
    return pos, d, dist, steps, kfirst


def sort_bins(points, d):
    """ray_setup_block's polar cells (16 rings x 256 sectors x above/below, far rings first)"""
    rings, sectors = 16, 256
    ringw = np.float32(MAP * RES * 0.5 / rings)
    fdx, fdy = d[:, 0].astype(np.float32), d[:, 1].astype(np.float32)
    ring = np.clip((np.sqrt(fdx * fdx + fdy * fdy) / ringw).astype(np.int64), 0, rings - 1)
    sec = np.clip(((np.arctan2(fdy, fdx) + np.float32(3.14159265)) * np.float32(sectors / 6.2831853)).astype(np.int64), 0, sectors - 1)
    hvz = np.floor(points[:, 2].astype(np.float32) / np.float32(RES)).astype(np.int64)
    return ((rings - 1 - ring) * 2 + (hvz >= 0)) * sectors + sec


def walk(pos, d, dist, k_lo, k_hi, chunk=4096):
    """per ray and step k in [k_lo, k_hi): emission flag, iter_steps (0: weight zero / no emission), mid.  Yields per chunk of rays."""
    n = d.shape[0]
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        kmax = int(k_hi[a:b].max())
        kmin = int(max(0, k_lo[a:b].min() - 1))
        ks = np.arange(kmin, kmax, dtype=np.int64)[None, :]
        ln = 1 + ks * HALF
        proj = [pos[c] + tdiv(d[a:b, c:c + 1] * ln, dist[a:b, None]) for c in range(3)]
        idx = [tdiv(p, np.int64(RES)) for p in proj]
        prevx = np.concatenate([np.zeros((b - a, 1), np.int64) if kmin == 0 else idx[0][:, :1] * 0 + tdiv(pos[0] + tdiv(d[a:b, 0:1] * (1 + (kmin - 1) * HALF), dist[a:b, None]), np.int64(RES)), idx[0][:, :-1]], axis=1)
        prevy = np.concatenate([np.zeros((b - a, 1), np.int64) if kmin == 0 else idx[1][:, :1] * 0 + tdiv(pos[1] + tdiv(d[a:b, 1:2] * (1 + (kmin - 1) * HALF), dist[a:b, None]), np.int64(RES)), idx[1][:, :-1]], axis=1)
        emit = (idx[0] != prevx) | (idx[1] != prevy)
        inr = (ks >= k_lo[a:b, None]) & (ks < k_hi[a:b, None])
        emit &= inr
        hit = pos[None, :] + d[a:b]
        dd = [hit[:, c:c + 1] - (idx[c] * RES + HALF) for c in range(3)]
        val = np.sqrt((dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]).astype(np.float32)).astype(np.int64)
        val = np.minimum(val, TAU)
        val = np.where(ln > dist[a:b, None], -val, val)
        wzero = (val < -(TAU // 10)) & (64 * (TAU + val) < TAU - TAU // 10)
        dz = (DZ * ln) >> 15
        it = np.where(dz * 2 >= RES, (dz * 2) // RES + 1, 1)
        mid = np.where(dz * 2 >= RES, dz // RES, 0)
        it = np.where(emit & ~wzero, it, 0)
        mid = np.broadcast_to(mid, it.shape)
        yield a, b, kmin, inr, emit, it, mid, val


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--alt", action="store_true")
    args = ap.parse_args()
    pts = S.os1_128_scan()
    pos, d, dist, steps, kfirst = ray_setup(pts)
    n = pts.shape[0]
    print(f"rays {n}, ray steps {int(steps.sum()) / 1e6:.2f} M (free-space part {int(kfirst.sum()) / 1e6:.2f} M, tails {int((steps - kfirst).sum()) / 1e6:.2f} M)")

    # ---------------- free pass ----------------
    zero = np.zeros(n, np.int64)
    lanes = 4
    ch = (kfirst + lanes - 1) // lanes
    samp_live = samp_slots = 0
    cand_total = 0
    dda_live = dda_slots = 0
    per_lane_emis = np.zeros((n, lanes), np.int64)
    for a, b, kmin, inr, emit, it, mid, val in walk(pos, d, dist, zero, kfirst):
        ks = np.arange(kmin, kmin + emit.shape[1])[None, :]
        for c in range(lanes):
            lo, hi = c * ch[a:b, None], np.minimum((c + 1) * ch[a:b, None], kfirst[a:b, None])
            per_lane_emis[a:b, c] = (emit & (ks >= lo) & (ks < hi)).sum(1)
        cand_total += int(emit.sum())
    # the sample k == 0 is outside the loop: lane 0 of a ray has ch - 1 iterations
    lane_steps = np.stack([np.clip(np.minimum((c + 1) * ch, kfirst) - c * ch, 0, None) for c in range(lanes)], 1)
    lane_iters = lane_steps.copy()
    lane_iters[:, 0] = np.clip(lane_iters[:, 0] - 1, 0, None)
    w = lane_iters.reshape(-1, 16 * lanes)  # a wave: 16 consecutive rays x 4 lanes
    n_iter = w.max(1)
    samp_live, samp_slots = int(w.sum()), int(n_iter.sum()) * 64
    e = per_lane_emis.reshape(-1, 16 * lanes)
    batches = (e.sum(1) + 63) // 64
    print("\nfree pass (64 rays x 4 lanes per workgroup)")
    print(f"  sample phase: {samp_live / 1e6:.2f} M samples in {samp_slots / 1e6:.2f} M lane-slots = {100.0 * samp_live / samp_slots:.1f} % useful, "
          f"{int(n_iter.sum()) / 1e3:.0f} k wave iterations")
    print(f"  emit phase  : {cand_total / 1e6:.2f} M candidates in {int(batches.sum()) / 1e3:.0f} k batches of 64 = {100.0 * cand_total / (64.0 * batches.sum()):.1f} % useful")
    if args.alt:
        it_dda = e.max(1)
        print(f"  [alt] one lane iteration per COLUMN CHANGE (no sample phase, no queue): {int(it_dda.sum()) / 1e3:.0f} k wave iterations, "
              f"{100.0 * e.sum() / (64.0 * it_dda.sum()):.1f} % useful")
        # work units of a fixed number of steps dealt out to the 256 lanes of the workgroup round-robin
        for unit in (32, 64, 128):
            tot_it = tot_live = 0
            em_cum = None
        # (evaluated below with the per-step emission table)

    # ---------------- tail march ----------------
    order = np.argsort(sort_bins(pts, d), kind="stable")
    parts = 8
    tl = steps - kfirst
    chp = (tl + parts - 1) // parts
    sample_live = sample_slots = 0
    emit_samples = emit_batches = 0
    round_live = round_slots = 0
    rounds_total = 0
    targets = 0
    alt_slots = 0
    po, do, disto, kf_o, st_o = pts[order], d[order], dist[order], kfirst[order], steps[order]
    chp_o = chp[order]
    for a, b, kmin, inr, emit, it, mid, val in walk(pos, do, disto, kf_o, st_o, chunk=64):
        # one work item pair: 64 rays; wave p walks [kf + p*ch, kf + (p+1)*ch)
        ks = np.arange(kmin, kmin + emit.shape[1])[None, :]
        for p in range(parts):
            lo = np.minimum(kf_o[a:b, None] + p * chp_o[a:b, None], st_o[a:b, None])
            hi = np.minimum(lo + chp_o[a:b, None], st_o[a:b, None])
            m = (ks >= lo) & (ks < hi)
            lane_n = m.sum(1)
            if lane_n.max() == 0:
                continue
            sample_live += int(lane_n.sum())
            sample_slots += int(lane_n.max()) * 64
            em = emit & m
            # queue order: sample iteration major, lane minor
            jj, ll = np.nonzero(em.T)  # iteration (relative), lane
            # iteration index relative to each lane's own start
            rel = (ks - lo)[em]
            order_q = np.lexsort((np.nonzero(em)[0], rel))
            its = it[em][order_q]
            mids = mid[em][order_q]
            nq = its.shape[0]
            emit_samples += nq
            for q0 in range(0, nq, 64):
                bi, bm = its[q0:q0 + 64], mids[q0:q0 + 64]
                emit_batches += 1
                if bi.max() == 0:
                    continue
                # round -1
                rounds_total += 1
                round_live += int((bi > 0).sum())
                round_slots += 64
                t_batch = int((bi > 0).sum())
                for r in range(int(bi.max())):
                    on = (r < bi) & (r != bm)
                    rounds_total += 1
                    round_live += int(on.sum())
                    round_slots += 64
                    t_batch += int(on.sum())
                targets += t_batch
                alt_slots += (t_batch + 63) // 64 * 64
    print("\ntail march (64 sorted rays x 8 parts, one part per wave)")
    print(f"  sample phase: {sample_live / 1e6:.2f} M samples in {sample_slots / 1e6:.2f} M lane-slots = {100.0 * sample_live / sample_slots:.1f} % useful")
    print(f"  emit phase  : {emit_samples / 1e6:.2f} M queued samples in {emit_batches / 1e3:.0f} k batches = {100.0 * emit_samples / (64.0 * emit_batches):.1f} % useful")
    print(f"  target rounds: {targets / 1e6:.2f} M scatter targets in {rounds_total / 1e3:.0f} k rounds = {100.0 * round_live / round_slots:.1f} % useful "
          f"({rounds_total / max(emit_batches, 1):.2f} rounds per batch)")
    if args.alt:
        print(f"  [alt] fan steps spread over the lanes (targets of a batch compacted): {alt_slots / 64e3:.0f} k rounds, {100.0 * targets / alt_slots:.1f} % useful")


if __name__ == "__main__":
    main()
