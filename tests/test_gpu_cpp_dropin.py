"""The C++ drop-in classes (include/warpsense_hip/{compat,mapping}.hpp) driven by examples/harness.cpp:
same call sequence as the reference's test/pcd_registration.cpp, result compared with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from warpsense_amd import synthetic as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_harness_matches_oracle(tmp_path):
    exe = os.path.join(ROOT, "examples", "harness")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    tau, res, mw, edge = 1000, 50, 640, 96
    pts = S.os1_128_scan(rings=32, azimuths=256, half_extents_mm=(2000.0, 1700.0, 800.0), seed=2)
    pert = S.transform_points_mm(pts, S.perturbation(35, -20, 8, 1.5))
    pts.tofile(tmp_path / "scan.bin")
    pert.tofile(tmp_path / "pert.bin")
    out = subprocess.run([exe, str(tmp_path / "scan.bin"), str(tmp_path / "pert.bin"), str(len(pts)), str(edge), str(res),
                          str(tau), str(mw), str(tmp_path / "avg.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    iters = int(lines[0].split()[1])
    T = np.array([float(x) for x in lines[1].split()], dtype=np.float32).reshape(4, 4).T

    oa = O.OracleMap((edge, edge, edge), tau, 0)
    on = oa.copy()
    O.update_tsdf(oa, on, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
    avg = np.fromfile(tmp_path / "avg.bin", dtype=np.uint32)
    assert np.array_equal(avg, oa.data)
    To, ito, _ = O.register_cloud(oa, pert, np.eye(4), 200, 0.1, 0.03, res)
    assert iters == ito
    assert np.abs(T - To).max() < 1e-4


def test_the_three_dropin_routes_agree(tmp_path):
    """examples/dropin_bench.cpp: the reference's callers unchanged (host vectors, one perform_registration + host solve per
    iteration -- tsdf_registration.cpp:55-92), the resident device loop behind the same host vectors, and device-resident
    clouds: same iteration count and bit-identical pose on all three (the binary exits 3 otherwise), and the iteration count
    is the oracle's."""
    import json
    exe = os.path.join(ROOT, "examples", "dropin_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    tau, res, mw, edge = 1000, 50, 640, 96
    pts = S.os1_128_scan(rings=32, azimuths=256, half_extents_mm=(2000.0, 1700.0, 800.0), seed=2)
    pert = S.transform_points_mm(pts, S.perturbation(35, -20, 8, 1.5))
    pts.tofile(tmp_path / "scan.bin")
    pert.tofile(tmp_path / "pert.bin")
    out = subprocess.run([exe, str(tmp_path / "scan.bin"), str(tmp_path / "pert.bin"), str(len(pts)), str(edge), str(res), str(tau), str(mw), "3"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["routes_agree"] is True
    # (every step applies the same scan again: the values of the map do not change after the first update -- the average of equal
    # values -- and the registration reads values and zero / non-zero weights only, so all registrations see the same map)
    oa = O.OracleMap((edge, edge, edge), tau, 0)
    on = oa.copy()
    O.update_tsdf(oa, on, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
    _, ito, _ = O.register_cloud(oa, pert, np.eye(4), 200, 0.1, 0.03, res)
    assert d["iterations"] == ito


@pytest.mark.parametrize("async_shift", [False, True])
def test_cpp_replay_matches_python_app(tmp_path, async_shift):
    """(async_shift: MappingNode::shift_map_async in C++ against the synchronous Python App — the map shift off the scan
    path must give the same poses, window and .h5 file.)
    examples/replay.cpp (warpsense::App, include/warpsense_hip/app.hpp: device pre-processing, update, registration,
    device-side map shift, export) against warpsense_amd.App on the same stream — itself checked against the
    oracle-driven sequence in tests/test_gpu_replay.py.  Same C ABI underneath, so everything must agree exactly."""
    import warpsense_amd as W
    from warpsense_amd import build
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "examples", "replay")
    tau, res, mw, edge, shift_m = 1000, 50, 640, 128, 0.6
    n_scans, rings, az = 6, 32, 256
    clouds = []
    for k in range(n_scans):
        sensor = np.array([k * 180.0, 0.5 * k * 180.0, 0.0])
        pts = S.os1_128_scan(sensor_mm=tuple(sensor), rings=rings, azimuths=az, half_extents_mm=(2600.0, 2200.0, 1100.0), seed=100 + k)
        clouds.append(((pts.astype(np.float64) - sensor) / 1000.0).astype(np.float32))
    np.stack(clouds).tofile(tmp_path / "clouds.bin")
    with_h5 = build.find_hdf5() is not None and build.build_h5() is not None
    args = [exe, str(tmp_path / "clouds.bin"), str(n_scans), str(rings * az), str(edge), str(res), str(tau), str(mw), str(shift_m),
            str(tmp_path / "poses.bin"), str(tmp_path / "map.bin")]
    if with_h5:
        args.append(str(tmp_path / "cpp.h5"))
    env = dict(os.environ)
    if async_shift:
        env["WS_REPLAY_ASYNC_SHIFT"] = "1"
    out = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = out.stdout.strip().splitlines()

    size = (edge, edge, edge // 2)
    params = W.Params(W.MapParams(resolution=res, max_distance=tau / 1000.0, max_weight=mw // 64, size=tuple(s * res / 1000.0 for s in size),
                                  shift=shift_m), W.RegistrationParams(200, 0.1, 0.03))
    app = W.App(params, str(tmp_path / "py.h5") if with_h5 else None)
    for c in clouds:
        app.cloud_callback(c)
    for k, t in enumerate(app.timings):
        f = lines[k].split()
        assert (int(f[1]), int(f[3]), int(f[5])) == (k, t["points"], t["iterations"]), (lines[k], t)
    last = lines[n_scans - 1].split()
    assert (int(last[7]), int(last[9])) == (app.n_updates, app.n_shifts) and app.n_shifts >= 1
    poses = np.fromfile(tmp_path / "poses.bin", dtype=np.float32).reshape(n_scans, 4, 4).transpose(0, 2, 1)
    assert np.array_equal(poses, np.stack(app.poses))
    lm = app.hdf5_local_map_
    w = [int(v) for v in lines[n_scans].split()[2:5]] + [int(v) for v in lines[n_scans].split()[6:9]]
    assert w == list(lm.pos) + list(lm.offset)
    host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
    app.gpu_.tsdf().avg_map().to_host(host)
    assert np.array_equal(np.fromfile(tmp_path / "map.bin", dtype=np.uint32), host.data_)
    app.terminate()
    if with_h5:
        a = W.GlobalMap(tau, 0, filename=str(tmp_path / "cpp.h5"), open_existing=True)
        b = W.GlobalMap(tau, 0, filename=str(tmp_path / "py.h5"), open_existing=True)
        import ctypes as C
        na, nb = C.c_int64(0), C.c_int64(0)
        a._H.ws_h5_num_chunks(a._file, C.byref(na))
        b._H.ws_h5_num_chunks(b._file, C.byref(nb))
        assert na.value == nb.value > 0
        pos = np.zeros((na.value, 3), dtype=np.int32)
        a._H.ws_h5_list_chunks(a._file, pos.ctypes.data_as(C.c_void_p), na.value, C.byref(na))
        for key in pos:
            assert np.array_equal(a.activate_chunk(*key).copy(), b.activate_chunk(*key)), key
        a._H.ws_h5_num_poses(a._file, C.byref(na))
        b._H.ws_h5_num_poses(b._file, C.byref(nb))
        assert na.value == nb.value == n_scans
        va, vb = np.zeros(7, np.float32), np.zeros(7, np.float32)
        for i in range(n_scans):
            a._H.ws_h5_read_pose(a._file, i, va.ctypes.data_as(C.c_void_p))
            b._H.ws_h5_read_pose(b._file, i, vb.ctypes.data_as(C.c_void_p))
            assert np.array_equal(va, vb)
        a.close()
        b.close()
