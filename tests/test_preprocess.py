"""Scan pre-processing: App::preprocess, src/warpsense/app.cpp:119-148.

CPU part: the C oracle against an independent numpy restatement of the same float/int arithmetic.
GPU part: ws_scan_preprocess* against the oracle — same points, same (first occurrence) order, bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from warpsense_amd import synthetic as S


def _wrap32(v: int) -> int:
    return (v + 2 ** 31) % 2 ** 32 - 2 ** 31


def numpy_preprocess(xyz, pose, res):
    """independent restatement: float32 arithmetic step by step, wrapping int32 products, a python set"""
    f = np.float32
    a = np.asarray(xyz, dtype=np.float32)[:, :3]
    M = (np.asarray(pose, dtype=np.float32) * f(32768)).astype(np.int32)  # to_int_mat
    out, seen = [], set()
    for x, y, z in a:
        if not (np.isfinite(x) and np.isfinite(y) and np.isfinite(z)):
            continue
        if float(x) < 0.3 and float(y) < 0.3 and float(z) < 0.3:
            continue
        c = [int(np.int32(f(f(np.floor(f(f(v * f(1000.0)) / f(res)))) * f(res)) + f(res // 2))) for v in (x, y, z)]
        q = []
        for r in range(3):
            acc = 0
            for k in range(3):
                acc = _wrap32(acc + _wrap32(int(M[r, k]) * c[k]))
            acc = _wrap32(acc + int(M[r, 3]))
            q.append(int(abs(acc) // 32768) * (1 if acc >= 0 else -1))  # C division truncates toward zero
        q = tuple(q)
        if q not in seen:
            seen.add(q)
            out.append(q)
    return np.array(out, dtype=np.int32).reshape(-1, 3)


def make_cloud(n, seed, stride=3):
    rng = np.random.default_rng(seed)
    a = np.zeros((n, stride), dtype=np.float32)
    a[:, :3] = rng.uniform(-12.0, 12.0, size=(n, 3)).astype(np.float32)
    k = n // 5
    a[:k, :3] = np.round(a[:k, :3] * 20) / 20          # many points on voxel borders (multiples of 50 mm)
    a[k:2 * k, :3] = a[:k, :3] + np.float32(0.004)      # near-duplicates: same voxel as another point
    a[2 * k:2 * k + 10, :3] = [0.1, 0.2, -0.4]          # dropped: all three below 0.3
    a[2 * k + 10:2 * k + 20, :3] = [0.1, 0.5, -3.0]     # kept: y >= 0.3
    if stride > 3:
        a[:, 3:] = rng.uniform(0, 255, size=(n, stride - 3))
    return a


POSES = [np.eye(4, dtype=np.float32), S.perturbation(1234.5, -987.25, 40.0, 17.0), S.perturbation(-20000.0, 15000.0, -800.0, -133.0)]


@pytest.mark.parametrize("res", [50, 64, 20])
def test_oracle_matches_numpy_restatement(res):
    cloud = make_cloud(3000, seed=res)
    for pose in POSES:
        got = O.preprocess(cloud, pose, res)
        want = numpy_preprocess(cloud, pose, res)
        assert np.array_equal(got, want)
        assert len({tuple(p) for p in got}) == len(got) > 1000


def test_oracle_snaps_to_voxel_centres_and_drops_near_points():
    pts = np.array([[1.0, 1.0, 1.0], [1.049, 1.0, 1.0], [1.05, 1.0, 1.0], [-1.0, 2.0, 2.0], [-0.001, 2.0, 2.0],
                    [0.29, 0.29, 0.29], [-5.0, -5.0, -5.0], [0.31, 0.0, 0.0]], dtype=np.float32)
    got = O.preprocess(pts, np.eye(4), 50)
    # 1.0 m and 1.049 m share a voxel; (-5,-5,-5) is dropped by the reference's signed test; 0.31 keeps the point
    assert got.tolist() == [[1025, 1025, 1025], [1075, 1025, 1025], [-975, 2025, 2025], [-25, 2025, 2025], [325, 25, 25]]


@pytest.mark.gpu
@pytest.mark.parametrize("res,stride,n", [(50, 3, 131072), (64, 4, 30000), (20, 3, 1), (50, 5, 257)])
def test_device_preprocess_matches_oracle(res, stride, n):
    import torch
    import warpsense_amd as W
    pre = W.ScanPreprocessor(max(n, 16))
    cloud = make_cloud(n, seed=n + res, stride=stride) if n > 1 else np.array([[3.0, -2.0, 1.0] + [0.0] * (stride - 3)], dtype=np.float32)
    for pose in POSES:
        want = O.preprocess(cloud, pose, res)
        got_host = pre.preprocess(cloud, pose, res).to_host()
        assert np.array_equal(got_host, want)
        dev = pre.preprocess(torch.from_numpy(cloud).cuda(), pose, res)
        assert len(dev) == len(want) and np.array_equal(dev.to_host(), want)


@pytest.mark.gpu
def test_device_preprocess_edge_cases():
    import warpsense_amd as W
    pre = W.ScanPreprocessor(1024)
    empty = pre.preprocess(np.zeros((0, 3), dtype=np.float32), np.eye(4), 50)
    assert len(empty) == 0 and empty.to_host().shape == (0, 3)
    near = np.full((100, 3), 0.1, dtype=np.float32)
    assert len(pre.preprocess(near, np.eye(4), 50)) == 0
    same = np.tile(np.array([[2.0, 2.0, 2.0]], dtype=np.float32), (1000, 1))
    assert pre.preprocess(same, np.eye(4), 50).to_host().tolist() == [[2025, 2025, 2025]]
    bad = np.array([[np.nan, 1, 1], [np.inf, 1, 1], [5, 5, 5]], dtype=np.float32)
    assert pre.preprocess(bad, np.eye(4), 50).to_host().tolist() == [[5025, 5025, 5025]]
    # beyond +-65.5 m the reference's int32 fixed-point product wraps: same (meaningless) integers as the restatement
    far = np.array([[2000.0, 70.0, -66.0], [65.0, -65.0, 1.0]], dtype=np.float32)
    assert np.array_equal(pre.preprocess(far, np.eye(4), 50).to_host(), O.preprocess(far, np.eye(4), 50))
    with pytest.raises(W.WsError):
        pre.preprocess(np.zeros((2000, 3), dtype=np.float32), np.eye(4), 50)  # more points than reserved


@pytest.mark.gpu
def test_preprocessed_points_feed_update_and_registration():
    """the device-resident output goes straight into update_tsdf / register_cloud: same map and pose as feeding the
    oracle's pre-processed points from the host."""
    import warpsense_amd as W
    tau, res, size = 1000, 50, (96, 96, 48)
    sensor = np.eye(4, dtype=np.float32)
    pts_mm = S.os1_128_scan(rings=32, azimuths=256, half_extents_mm=(2000.0, 1800.0, 900.0), seed=3)
    cloud = (pts_mm.astype(np.float32) / np.float32(1000.0))
    params = W.Params(W.MapParams(resolution=res, max_distance=1.0, max_weight=10, size=tuple(s * res / 1000.0 for s in size)))
    out = []
    for mode in ("device", "host"):
        lm = W.LocalMap(*size, tau, 0)
        reg = W.TSDFRegistration(params, lm)
        if mode == "device":
            scan = W.ScanPreprocessor().preprocess(cloud, sensor, res)
        else:
            scan = O.preprocess(cloud, sensor, res)
        reg.update_tsdf(scan, pose=sensor)
        T = reg.register_cloud(scan, S.perturbation(20, -15, 5, 1.0))
        host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
        reg.tsdf().avg_map().to_host(host)
        out.append((host.data_.copy(), T, reg.last_iterations))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
