"""Device-side map shift against the host mirror of HDF5LocalMap::shift, which is itself pinned
by the reference's map_raw test (tests/test_oracle_pins.py::test_kat_ring_buffer_shift)."""
import numpy as np
import pytest

import oracle_lib as O
from warpsense_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _maps(size, tau=1000, seed=0):
    import warpsense_amd as W
    rng = np.random.default_rng(seed)
    dev_lm = W.LocalMap(*size, tau, 0)
    n = dev_lm.data.size
    dev_lm.data[:] = W.pack_entry(rng.integers(-tau, tau + 1, n), rng.integers(-64, 641, n))
    host_lm = W.LocalMap(*size, tau, 0)
    host_lm.data[:] = dev_lm.data
    params = W.Params(W.MapParams(resolution=50, max_distance=tau / 1000.0, max_weight=10, size=tuple(s * 0.05 for s in size)))
    tm = W.TSDFMapping(params, dev_lm)
    # make new_map the default map again (TSDFCuda copies the host map into both device maps)
    blank = W.LocalMap(*size, tau, 0)
    tm.tsdf().new_map().to_device(blank.device_map())
    return W, tm, dev_lm, host_lm


def _download(W, tm, lm, which=0):
    host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
    (tm.tsdf().avg_map() if which == 0 else tm.tsdf().new_map()).to_host(host)
    return host


def test_shift_sequence_matches_host_mirror():
    W, tm, dev_lm, host_lm = _maps((21, 17, 13))
    for new_pos in [(3, 0, 0), (3, -4, 2), (10, -4, 2), (10, 5, -3), (-2, 5, -3), (0, 0, 0)]:
        tm.shift_map(new_pos)
        host_lm.shift(new_pos)
        got = _download(W, tm, dev_lm)
        assert list(got.pos_) == list(host_lm.pos) and list(got.offset_) == list(host_lm.offset)
        assert np.array_equal(got.data_, host_lm.data), new_pos
    # coming back to the origin restores every voxel (nothing was lost in the global store)
    W2, tm2, dev2, ref = _maps((21, 17, 13))
    assert np.array_equal(_download(W, tm, dev_lm).data_, ref.data)


def test_box_roundtrip_and_bounds():
    W, tm, dev_lm, host_lm = _maps((15, 15, 15), seed=3)
    avg = tm.tsdf().avg_map()
    lo, hi = (-7, -2, 3), (1, 4, 7)
    box = avg.extract_box(lo, hi)
    want = np.array([host_lm.data[host_lm.get_index(x, y, z)] for x in range(lo[0], hi[0] + 1) for y in range(lo[1], hi[1] + 1)
                     for z in range(lo[2], hi[2] + 1)], dtype=np.uint32)
    assert np.array_equal(box, want)
    avg.insert_box(lo, hi, box[::-1].copy())
    assert np.array_equal(avg.extract_box(lo, hi), box[::-1])
    with pytest.raises(W.WsError):
        avg.extract_box((-8, 0, 0), (0, 0, 0))  # outside the window


def test_update_after_shift_matches_oracle():
    """a TSDF update on the shifted window == oracle on a map with the same pos/offset."""
    import torch
    import warpsense_amd as W
    tau, res, mw, size = 1000, 50, 640, (64, 64, 32)
    lm = W.LocalMap(*size, tau, 0)
    params = W.Params(W.MapParams(resolution=res, max_distance=1.0, max_weight=10, size=tuple(s * res / 1000.0 for s in size)))
    tm = W.TSDFMapping(params, lm)
    tm.shift_map((5, -3, 2))
    sensor = (5 * res + 10, -3 * res + 7, 2 * res + 3)
    pts = S.os1_128_scan(sensor_mm=sensor, rings=16, azimuths=128, half_extents_mm=(1400.0, 1300.0, 700.0), seed=4)
    pos = [int(np.floor(np.float32(s) / np.float32(res))) for s in sensor]
    tm.update_tsdf(torch.from_numpy(pts).cuda(), pos_rm=pos, up_rm=(0, 0, 32768))
    got = _download(W, tm, lm)
    oa = O.OracleMap(size, tau, 0, pos=lm.pos, offset=lm.offset)
    on = oa.copy()
    O.update_tsdf(oa, on, pts, pos, (0, 0, 32768), tau, mw, res)
    assert np.array_equal(got.data_, oa.data)


def test_write_back_exports_device_map_to_h5(tmp_path):
    """TSDFMapping.write_back: chunks gathered from the device ring buffer and written to the .h5 file
    == the host path of the reference (whole-map download, HDF5LocalMap::write_back voxel by voxel), after an update
    on a shifted window so the ring-buffer offsets are non-trivial."""
    import torch
    from warpsense_amd import build
    if build.find_hdf5() is None or build.build_h5() is None:
        pytest.skip("no HDF5 C library on this box")
    import warpsense_amd as W
    tau, res, size = 1000, 50, (96, 80, 72)
    path = str(tmp_path / "export.h5")
    mp = W.MapParams(resolution=res, max_distance=1.0, max_weight=10, size=tuple(s * res / 1000.0 for s in size))
    g = W.GlobalMap(tau, 0, filename=path, map_params=mp)
    lm = W.LocalMap(*size, tau, 0, g)
    tm = W.TSDFMapping(W.Params(mp), lm)
    tm.shift_map((37, -50, 11))  # window straddles chunk borders in every axis, negative chunk ids in y
    sensor = (37 * res + 10, -50 * res + 7, 11 * res + 3)
    pts = S.os1_128_scan(sensor_mm=sensor, rings=32, azimuths=256, half_extents_mm=(2100.0, 1700.0, 1500.0), seed=9)
    pos = [int(np.floor(np.float32(s) / np.float32(res))) for s in sensor]
    tm.update_tsdf(torch.from_numpy(pts).cuda(), pos_rm=pos, up_rm=(0, 0, 32768))
    tm.write_back()
    g.write_pose(np.eye(4), 1000.0)
    g.close()

    # the reference's route: download the window, save it voxel-for-voxel into a (memory-only) global map
    host = _download(W, tm, lm)
    ref_g = W.GlobalMap(tau, 0)
    ref_lm = W.LocalMap(*size, tau, 0, ref_g)
    ref_lm.pos[:], ref_lm.offset[:] = host.pos_, host.offset_
    ref_lm.data[:] = host.data_
    ref_lm.write_back()
    assert np.count_nonzero(W.unpack_entry(host.data_)[1]) > 10_000  # the scan is in there

    g2 = W.GlobalMap(tau, 0, filename=path, open_existing=True)
    assert len(ref_g.chunks) >= 8
    for key, want in ref_g.chunks.items():
        assert np.array_equal(g2.activate_chunk(*key), want), key
    # the file also holds the chunks the (still blank) slabs went to when the window was shifted: default entries only
    import ctypes as C
    n = C.c_int64(0)
    g2._H.ws_h5_num_chunks(g2._file, C.byref(n))
    pos_list = np.zeros((n.value, 3), dtype=np.int32)
    g2._H.ws_h5_list_chunks(g2._file, pos_list.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
    in_file = {tuple(int(v) for v in p) for p in pos_list}
    assert set(ref_g.chunks) <= in_file
    for key in in_file - set(ref_g.chunks):
        assert np.all(g2.activate_chunk(*key) == W.pack_entry(tau, 0)), key
    g2.close()


def test_async_shift_equals_synchronous_shift():
    """TSDFMapping.shift_map_async (window moved by device kernels inside the call, leaving slabs filed by a worker
    thread) == shift_map, for a path that leaves a region and comes back to it (revisited chunks are uploaded again),
    with diagonal moves (a corner that enters with x leaves again with y in the same shift)."""
    W, tm_a, dev_a, host_lm = _maps((21, 17, 13), seed=5)
    _, tm_s, dev_s, _ = _maps((21, 17, 13), seed=5)
    for new_pos in [(3, 0, 0), (3, -4, 2), (10, 5, -3), (-2, -6, 4), (0, 0, 0), (7, 7, 7), (0, 0, 0)]:
        tm_a.shift_map_async(new_pos)
        tm_s.shift_map(new_pos)
        host_lm.shift(new_pos)
        got_a, got_s = _download(W, tm_a, dev_a), _download(W, tm_s, dev_s)
        assert list(got_a.pos_) == list(host_lm.pos) and list(got_a.offset_) == list(host_lm.offset)
        assert np.array_equal(got_s.data_, host_lm.data), new_pos
        assert np.array_equal(got_a.data_, host_lm.data), new_pos
    tm_a.wait_shift()
    # both global maps hold the same chunks
    ga, gs = dev_a.map_, dev_s.map_
    assert set(ga.chunks) == set(gs.chunks)
    for key in gs.chunks:
        assert np.array_equal(ga.chunks[key], gs.chunks[key]), key


def test_async_shift_does_not_stall_the_stream():
    """a scan enqueued right after shift_map_async runs against the moved window and gives the same map as after the
    synchronous shift"""
    import torch
    import warpsense_amd as W
    tau, res, mw, size = 1000, 50, 640, (64, 64, 32)
    outs = []
    for asyn in (False, True):
        lm = W.LocalMap(*size, tau, 0)
        params = W.Params(W.MapParams(resolution=res, max_distance=1.0, max_weight=10, size=tuple(s * res / 1000.0 for s in size)))
        tm = W.TSDFMapping(params, lm)
        for k, pos in enumerate([(0, 0, 0), (5, -3, 2), (9, 1, 2)]):
            if k:
                (tm.shift_map_async if asyn else tm.shift_map)(pos)
            sensor = (pos[0] * res + 10, pos[1] * res + 7, pos[2] * res + 3)
            pts = S.os1_128_scan(sensor_mm=sensor, rings=16, azimuths=128, half_extents_mm=(1400.0, 1300.0, 700.0), seed=4 + k)
            pts = (pts.astype(np.int64)).astype(np.int32)
            tm.update_tsdf(torch.from_numpy(pts).cuda(), pos_rm=list(pos), up_rm=(0, 0, 32768))
        tm.wait_shift()
        outs.append(_download(W, tm, lm).data_.copy())
    assert np.array_equal(outs[0], outs[1])
    assert int(np.count_nonzero(outs[0] != W.pack_entry(tau, 0))) > 10_000
