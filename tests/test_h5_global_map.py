"""Global-map file: libwarpsense_h5.so writes the layout of HDF5GlobalMap
(src/map/hdf5_global_map.cpp) — checked by reading the files back with independent HDF5 tools (h5dump, and h5py
of the conda python when the image has them) and by mirroring the reference's own global-map tests
(test/map.cpp:92-238).  CPU only; skipped where no HDF5 C library is installed."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from warpsense_amd import _lib, build

pytestmark = pytest.mark.skipif(build.find_hdf5() is None, reason="no HDF5 C library on this box")


@pytest.fixture(scope="module", autouse=True)
def _built():
    assert build.build_h5() is not None
    L = _lib.load_h5()
    for name in _lib.H5_EXPORTS:  # every symbol include/warpsense_h5.h declares
        assert hasattr(L, name), name


def _h5dump():
    for cand in (shutil.which("h5dump"), "/opt/conda/bin/h5dump"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _conda_h5py():
    py = "/opt/conda/bin/python"
    if not os.path.exists(py):
        return None
    r = subprocess.run([py, "-c", "import h5py"], capture_output=True)
    return py if r.returncode == 0 else None


def test_header_and_exports_agree():
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "warpsense_h5.h")).read()
    declared = set(re.findall(r"\b(ws_h5_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.H5_EXPORTS)


def test_layout_matches_reference_file_format(tmp_path):
    """Object names, types and shapes as HighFive writes them: /map/<cx>_<cy>_<cz> uint32[262144],
    /poses/<i>/pose float32[7], scalar int32/float32 attributes on /map."""
    import warpsense_amd as W
    path = str(tmp_path / "layout.h5")
    mp = W.MapParams(resolution=50, max_distance=1.0, max_weight=10, size=(25.6, 25.6, 25.6))
    g = W.GlobalMap(mp.tau, 0, filename=path, map_params=mp)
    g.set_value(1, 2, 3, 777, 5)
    g.set_value(-1, -64, -65, -123, 64)  # chunk (-1, -1, -2): floor division for negative coordinates
    T = np.eye(4)
    T[:3, 3] = (1234.5678, -2.0004, 0.0005)
    vals = g.write_pose(T, 1000.0)
    g.close()
    assert np.allclose(vals, [1.235, -0.002, 0.0, 0, 0, 0, 1], atol=1e-6)

    dump = _h5dump()
    if dump:
        out = subprocess.run([dump, "-H", path], capture_output=True, text=True, check=True).stdout
        assert 'DATASET "0_0_0"' in out and 'DATASET "-1_-1_-2"' in out
        assert "H5T_STD_U32LE" in out and "( 262144 ) / ( 262144 )" in out
        assert 'GROUP "poses"' in out and 'DATASET "pose"' in out and "H5T_IEEE_F32LE" in out and "( 7 ) / ( 7 )" in out
        for attr in ("tau", "map_size_x", "map_size_y", "map_size_z", "max_distance", "map_resolution", "max_weight"):
            assert f'ATTRIBUTE "{attr}"' in out
    py = _conda_h5py()
    if py:
        code = ("import h5py, json, sys; f = h5py.File(sys.argv[1], 'r'); c = f['/map/0_0_0'][...]; d = f['/map/-1_-1_-2'][...];"
                "print(json.dumps({'dtype': str(c.dtype), 'shape': list(c.shape), 'v': int(c[1*4096 + 2*64 + 3]),"
                "'w': int(d[63*4096 + 0*64 + 63]), 'fill': int(c[0]), 'pose': [float(x) for x in f['/poses/0/pose'][...]],"
                "'tau': int(f['/map'].attrs['tau']), 'res': int(f['/map'].attrs['map_resolution']),"
                "'md': float(f['/map'].attrs['max_distance']), 'sx': int(f['/map'].attrs['map_size_x']),"
                "'mw': int(f['/map'].attrs['max_weight'])}))")
        r = json.loads(subprocess.run([py, "-c", code, path], capture_output=True, text=True, check=True).stdout)
        assert r["dtype"] == "uint32" and r["shape"] == [262144]
        assert r["v"] == int(W.pack_entry(777, 5)) and r["w"] == int(W.pack_entry(-123, 64)) and r["fill"] == int(W.pack_entry(mp.tau, 0))
        assert np.allclose(r["pose"], vals)
        assert (r["tau"], r["res"], r["sx"], r["mw"]) == (1000, 50, 512, 640) and abs(r["md"] - 1.0) < 1e-7


def test_chunk_cache_evicts_to_file_and_reads_back(tmp_path):
    """More than NUM_CHUNKS = 64 active chunks: the oldest ones go to the file (hdf5_global_map.cpp:96-121) and
    come back unchanged; a re-opened file serves them too."""
    import warpsense_amd as W
    path = str(tmp_path / "lru.h5")
    g = W.GlobalMap(3000, 0, filename=path)
    rng = np.random.default_rng(5)
    coords = [(int(x), int(y), int(z)) for x, y, z in rng.integers(-400, 400, size=(300, 3))]
    expect = {}
    for i, (x, y, z) in enumerate(coords):
        g.set_value(x, y, z, -1000 + i, 1 + i % 60)
        expect[(x, y, z)] = (-1000 + i, 1 + i % 60)
    assert len(g.chunks) <= W.GlobalMap.NUM_CHUNKS
    for (x, y, z), vw in expect.items():
        assert g.get_value(x, y, z) == vw
    assert g.get_value(10_000, 0, 0) == (3000, 0)  # never written: default entry
    g.close()
    g2 = W.GlobalMap(3000, 0, filename=path, open_existing=True)
    for (x, y, z), vw in expect.items():
        assert g2.get_value(x, y, z) == vw
    H = _lib.load_h5()
    import ctypes as C
    n = C.c_int64(0)
    assert H.ws_h5_num_chunks(g2._file, C.byref(n)) == 0
    chunks = {(x // 64, y // 64, z // 64) for x, y, z in coords} | {(10_000 // 64, 0, 0)}  # a chunk that was only read is active too
    assert n.value == len(chunks)
    pos = np.zeros((n.value, 3), dtype=np.int32)
    assert H.ws_h5_list_chunks(g2._file, pos.ctypes.data_as(C.c_void_p), n.value, C.byref(n)) == 0
    assert {tuple(int(v) for v in p) for p in pos} == chunks
    g2.close()


def test_local_map_write_back_roundtrip_like_reference_test(tmp_path):
    """test/map.cpp:9-238 in miniature: fill a local map, write_back to the file, re-open, rebuild a second local map
    from the chunk datasets (cells with weight > 0 only) and compare; poses accumulate as /poses/0, /poses/1, ..."""
    import warpsense_amd as W
    path = str(tmp_path / "test.h5")
    tau = 3000
    g = W.GlobalMap(tau, 0, filename=path)
    lm = W.LocalMap(20, 20, 20, tau, 0, g)
    rng = np.random.default_rng(11)
    cells = {}
    for _ in range(400):
        x, y, z = (int(v) for v in rng.integers(-10, 11, size=3))
        v, w = int(rng.integers(-tau, tau + 1)), int(rng.integers(1, 640))
        lm.set_value(x, y, z, v, w)
        cells[(x, y, z)] = (v, w)
    lm.write_back()
    rot = np.eye(4)
    th = np.pi / 4
    rot[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    pose = np.eye(4)
    for _ in range(3):
        pose = pose @ rot          # target_pose.rotate(rotation)
        pose[:3, 3] += 1.0         # pretranslate(ones)
        g.write_pose(pose, 1.0)
    g.close()

    H = _lib.load_h5()
    import ctypes as C
    h = C.c_void_p()
    assert H.ws_h5_open(path.encode(), 0, C.byref(h)) == 0
    n = C.c_int64(0)
    assert H.ws_h5_num_poses(h, C.byref(n)) == 0 and n.value == 3
    from scipy.spatial.transform import Rotation
    target = np.eye(4)
    for i in range(3):
        vals = np.zeros(7, dtype=np.float32)
        assert H.ws_h5_read_pose(h, i, vals.ctypes.data_as(C.c_void_p)) == 0
        target = target @ rot
        target[:3, 3] += 1.0
        assert np.allclose(vals[:3], target[:3, 3], atol=5e-4)
        q = Rotation.from_matrix(target[:3, :3]).as_quat()  # x y z w
        assert np.allclose(vals[3:], q, atol=1e-3) or np.allclose(vals[3:], -q, atol=1e-3)
    n = C.c_int64(0)
    H.ws_h5_num_chunks(h, C.byref(n))
    pos = np.zeros((n.value, 3), dtype=np.int32)
    H.ws_h5_list_chunks(h, pos.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
    lm2 = W.LocalMap(20, 20, 20, tau, 0)
    buf = np.zeros(64 ** 3, dtype=np.uint32)
    ex = C.c_int32(0)
    for cx, cy, cz in pos:
        assert H.ws_h5_read_chunk(h, int(cx), int(cy), int(cz), buf.ctypes.data_as(C.c_void_p), C.byref(ex)) == 0 and ex.value == 1
        v, w = W.unpack_entry(buf)
        for idx in np.nonzero(w > 0)[0]:
            i, j, k = idx // 4096, (idx // 64) % 64, idx % 64
            lm2.set_value(64 * int(cx) + int(i), 64 * int(cy) + int(j), 64 * int(cz) + int(k), int(v[idx]), int(w[idx]))
    H.ws_h5_close(h)
    for (x, y, z), vw in cells.items():
        assert lm2.value(x, y, z) == vw
    assert lm2.value(10, 10, -10) == cells.get((10, 10, -10), (tau, 0))


def test_pose_values_follow_eigen_quaternion_branches():
    import warpsense_amd as W
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    mats = [Rotation.random(random_state=int(s)).as_matrix() for s in rng.integers(0, 1 << 30, size=40)]
    mats += [np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0])]  # trace <= 0 branches
    for R in mats:
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = rng.uniform(-5e4, 5e4, size=3)
        vals = W.pose_to_values(T, 1000.0)
        assert np.allclose(vals[:3], np.round(T[:3, 3] / 1000.0, 3), atol=1.1e-3)
        q = Rotation.from_matrix(R).as_quat()
        assert np.allclose(vals[3:], q, atol=1.1e-3) or np.allclose(vals[3:], -q, atol=1.1e-3)
        assert np.allclose(vals * 1000.0, np.round(vals * 1000.0), atol=1e-2)  # 3 decimals
