"""Synthetic OS1-128 scans of a box room — the workload BASELINE.json's metric is quoted on.

128 rings x 1024 azimuths = 131 072 rays in ring-major order (index = ring*1024 + azimuth); elevation
-22.5deg .. +22.5deg, sensor at `sensor_mm` inside an axis-aligned room with half extents
(10 000, 8 000, 2 500) mm centred on the map origin; range = ray/box intersection plus integer noise
((s >> 8) % 21) - 10 mm from the LCG s <- s*1664525 + 1013904223 (mod 2^32), seed 12345, one draw per ray;
point = trunc(d * dir) per axis, in integer millimetres (the hot path's unit, include/warpsense/consts.h).

Pure numpy; used by bench.py, __graft_entry__.smoke() and the tests.  No reference code involved.
"""
from __future__ import annotations

import numpy as np

RINGS = 128
AZIMUTHS = 1024
HALF_EXTENTS_MM = (10_000.0, 8_000.0, 2_500.0)


def lcg_noise(n: int, seed: int = 12345) -> np.ndarray:
    """((s >> 8) % 21) - 10 for n successive LCG states (uint32 arithmetic)."""
    # s_k = a^k s_0 + c (a^k - 1)/(a - 1) evaluated iteratively in blocks to stay vectorised
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    mask = np.uint64(0xFFFFFFFF)
    out = np.empty(n, dtype=np.int64)
    s = np.uint64(seed)
    # plain loop in chunks; 131 072 iterations of python are ~50 ms with this jump-ahead trick:
    # precompute (A_j, C_j) for j = 1..B so that s_{k+j} = A_j s_k + C_j
    B = 4096
    A = np.empty(B, dtype=np.uint64)
    C = np.empty(B, dtype=np.uint64)
    aj, cj = np.uint64(1), np.uint64(0)
    for j in range(B):
        aj = (aj * a) & mask
        cj = (cj * a + c) & mask
        A[j], C[j] = aj, cj
    k = 0
    while k < n:
        m = min(B, n - k)
        block = (A[:m] * s + C[:m]) & mask
        out[k:k + m] = ((block >> np.uint64(8)) % np.uint64(21)).astype(np.int64) - 10
        s = block[m - 1]
        k += m
    return out


def os1_128_scan(sensor_mm=(0.0, 0.0, 0.0), rings: int = RINGS, azimuths: int = AZIMUTHS, seed: int = 12345,
                 half_extents_mm=HALF_EXTENTS_MM, yaw_rad: float = 0.0) -> np.ndarray:
    """Return an (rings*azimuths, 3) int32 array of points in the MAP frame (mm)."""
    i = np.arange(rings, dtype=np.float64)
    j = np.arange(azimuths, dtype=np.float64)
    el = np.deg2rad(-22.5 + 45.0 * i / max(rings - 1, 1))
    az = 2.0 * np.pi * j / azimuths + yaw_rad
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    ca, sa = np.cos(az)[None, :], np.sin(az)[None, :]
    d = np.stack([ce * ca, ce * sa, np.broadcast_to(se, (rings, azimuths))], axis=-1).reshape(-1, 3)
    o = np.asarray(sensor_mm, dtype=np.float64)
    he = np.asarray(half_extents_mm, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_hi = (he - o) / d
        t_lo = (-he - o) / d
    t = np.where(d > 0, t_hi, np.where(d < 0, t_lo, np.inf)).min(axis=1)
    t = t + lcg_noise(rings * azimuths, seed).astype(np.float64)
    pts = o[None, :] + t[:, None] * d
    return np.trunc(pts).astype(np.int32)


def perturbation(tx_mm: float = 100.0, ty_mm: float = 100.0, tz_mm: float = 0.0, rz_deg: float = 5.0) -> np.ndarray:
    """The known SE(3) of test/pcd_registration.cpp:30-34 as a 4x4 float32 (row-major numpy) matrix."""
    a = np.deg2rad(rz_deg)
    T = np.eye(4, dtype=np.float64)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    T[:3, 3] = (tx_mm, ty_mm, tz_mm)
    return T.astype(np.float32)


def transform_points_mm(points: np.ndarray, T: np.ndarray) -> np.ndarray:
    """Round-half-away transform of integer mm points (include/util/util.h:80-101 semantics)."""
    p = points.astype(np.float32)
    Tf = T.astype(np.float32)
    q = p @ Tf[:3, :3].T + Tf[:3, 3]
    q = np.where(q < 0, q - np.float32(0.5), q + np.float32(0.5))
    return q.astype(np.int32)
