"""ROS-free replay node: warpsense_amd.App reproduces the sequencing of App::cloud_callback
(src/warpsense/app.cpp:65-117) — checked against the same sequence driven through the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O
from warpsense_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _angle(Ra, Rb):
    R = Ra.astype(np.float64).T @ Rb.astype(np.float64)
    return float(np.arctan2(np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2.0, (np.trace(R) - 1.0) / 2.0))


def sensor_clouds(n_scans, step_mm):
    """a sensor moving along +x through a box room; clouds in the SENSOR frame (metres)"""
    clouds = []
    for k in range(n_scans):
        sensor = np.array([k * step_mm, 0.5 * k * step_mm, 0.0])
        pts = S.os1_128_scan(sensor_mm=tuple(sensor), rings=32, azimuths=256, half_extents_mm=(2600.0, 2200.0, 1100.0), seed=100 + k)
        clouds.append(((pts.astype(np.float64) - sensor) / 1000.0).astype(np.float32))
    return clouds


def oracle_replay(clouds, size, tau, mw, res, reg, shift_m):
    """cloud_callback + map_shift with every device step replaced by the oracle"""
    import warpsense_amd as W
    lm = W.LocalMap(*size, tau, 0)
    om = O.OracleMap(size, tau, 0)
    on = om.copy()
    pose = np.eye(4, dtype=np.float32)
    last_tsdf, last_shift = pose.copy(), pose.copy()
    initialized = shifted = False
    poses, its, updates, shifts = [], [], 0, 0
    for cloud in clouds:
        scan = O.preprocess(cloud, pose, res)
        d = np.linalg.norm(last_tsdf[:3, 3] / np.float32(1000) - pose[:3, 3] / np.float32(1000))
        if not initialized or d > 0.3 or shifted:
            initialized, last_tsdf, shifted = True, pose.copy(), False
            pos, up = W.to_map(pose, res), W.to_int_mat(pose)[:3, 2]
            O.update_tsdf(om, on, scan, pos, up, tau, mw, res)
            updates += 1
        T, it, _ = O.register_cloud(om, scan, np.eye(4), reg[0], reg[1], reg[2], res)
        T = T.astype(np.float32)
        # app.cpp:172-176 in float32, products summed in index order -- spelled out: numpy's matmul goes through BLAS, whose
        # summation order / FMA use is not specified (it differed from the index-order sum by one ulp on the GPU box, which
        # the bit-exact pose assertion below caught once it stopped being a silent skip)
        R = np.zeros((3, 3), dtype=np.float32)
        for i in range(3):
            for j in range(3):
                acc = np.float32(0)
                for k in range(3):
                    acc = np.float32(acc + np.float32(T[i, k] * pose[k, j]))
                R[i, j] = acc
        pose[:3, :3] = R
        pose[:3, 3] += T[:3, 3]
        poses.append(pose.copy())
        its.append(it)
        if np.linalg.norm(last_shift[:3, 3] / np.float32(1000) - pose[:3, 3] / np.float32(1000)) >= shift_m:
            last_shift = pose.copy()
            lm.data[:] = om.data
            lm.pos[:], lm.offset[:] = om.pos, om.offset
            lm.shift(W.to_map(pose, res))
            om = O.OracleMap(size, tau, 0, pos=lm.pos, offset=lm.offset)
            om.data[:] = lm.data
            on = O.OracleMap(size, tau, 0, pos=lm.pos, offset=lm.offset)
            shifted = True
            shifts += 1
    return poses, its, updates, shifts, om


def test_replay_matches_oracle_sequence(tmp_path):
    import warpsense_amd as W
    from warpsense_amd import build
    tau, res, mw, size = 1000, 50, 640, (128, 128, 64)
    reg = (200, 0.1, 0.03)
    shift_m = 0.6
    h5 = str(tmp_path / "replay.h5") if build.find_hdf5() is not None and build.build_h5() else None
    params = W.Params(W.MapParams(resolution=res, max_distance=tau / 1000.0, max_weight=mw // 64, size=tuple(s * res / 1000.0 for s in size),
                                  shift=shift_m), W.RegistrationParams(*reg))
    clouds = sensor_clouds(6, 180.0)
    app = W.App(params, h5)
    for c in clouds:
        app.cloud_callback(c)
    want_poses, want_its, want_updates, want_shifts, om = oracle_replay(clouds, app.hdf5_local_map_.size, tau, mw, res, reg, shift_m)
    assert app.n_updates == want_updates >= 2 and app.n_shifts == want_shifts >= 1
    assert [t["iterations"] for t in app.timings] == want_its
    for got, want in zip(app.poses, want_poses):
        assert np.linalg.norm(got[:3, 3] - want[:3, 3]) / 1000.0 < 1e-4
        assert _angle(got[:3, :3], want[:3, :3]) < 1e-4
    # the sensor really moved and the estimate follows it (180 mm per scan along x, 90 mm along y; scan-to-map
    # registration against a map that is only refreshed every 0.3 m lags behind a little)
    assert abs(app.poses[-1][0, 3] - 5 * 180.0) < 200.0 and abs(app.poses[-1][1, 3] - 5 * 90.0) < 100.0
    # bit-identical poses (hard: a one-ulp difference must not skip the map comparison) -> identical final window
    for k, (got, want) in enumerate(zip(app.poses, want_poses)):
        assert np.array_equal(got, want), (k, np.abs(got - want).max())
    lm = app.hdf5_local_map_
    host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
    app.gpu_.tsdf().avg_map().to_host(host)
    assert np.array_equal(host.data_, om.data)
    app.terminate()
    if h5:
        g = W.GlobalMap(tau, 0, filename=h5, open_existing=True)
        import ctypes as C
        n = C.c_int64(0)
        g._H.ws_h5_num_poses(g._file, C.byref(n))
        assert n.value == len(clouds)
        vals = np.zeros(7, dtype=np.float32)
        g._H.ws_h5_read_pose(g._file, len(clouds) - 1, vals.ctypes.data_as(C.c_void_p))
        assert np.allclose(vals[:3], app.poses[-1][:3, 3] / 1000.0, atol=6e-4)
        g.close()
