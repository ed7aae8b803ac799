"""GPU parity of the Point-to-TSDF registration against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import oracle_lib as O
from warpsense_amd import synthetic as S

pytestmark = pytest.mark.gpu


def build_scene(size=(128, 128, 64), tau=1000, res=50, mw=640, rings=64, az=512, he=(2600.0, 2300.0, 1000.0), scans=1):
    import torch
    import warpsense_amd as W
    params = W.Params(W.MapParams(resolution=res, max_distance=tau / 1000.0, max_weight=mw // 64,
                                  size=tuple(s * res / 1000.0 for s in size)))
    lm = W.LocalMap(size[0], size[1], size[2], tau, 0)
    reg = W.TSDFRegistration(params, lm)
    oa = O.OracleMap(size, tau, 0)
    on = oa.copy()
    pts = None
    for k in range(scans):
        pts = S.os1_128_scan(rings=rings, azimuths=az, half_extents_mm=he, seed=21 + k)
        O.update_tsdf(oa, on, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
        reg.update_tsdf(torch.from_numpy(pts).cuda(), pose=np.eye(4, dtype=np.float32))
    return reg, oa, pts, res


def pose_error(A, B):
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    dt = np.linalg.norm(A[:3, 3] - B[:3, 3]) / 1000.0  # mm -> m
    R = A[:3, :3] @ B[:3, :3].T
    # atan2 form: well conditioned near 0 (arccos of a float32-rounded trace is not)
    k = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    ang = np.arctan2(np.linalg.norm(k), (np.trace(R) - 1) / 2)
    return dt, ang


@pytest.mark.parametrize("flags", [0, 1])
def test_hgec_bit_exact(flags):
    """h, g, e, c of one perform_registration call == oracle, for identity and a perturbed transform."""
    import warpsense_amd as W
    reg, oa, pts, res = build_scene()
    reg.reg_.flags = flags
    q = S.transform_points_mm(pts, S.perturbation(40, -30, 10, 2.0))
    reg.reg_.prepare_registration(q)
    for T in (np.eye(4, dtype=np.float32), S.perturbation(-35, 28, -9, -1.7), S.perturbation(500.5, 100.25, 3.75, 11.0)):
        h, g, e, c = reg.reg_.perform_registration(reg.tsdf().device_map(), T, res)
        ho, go, eo, co = O.reg_iterate(oa, T, q, res, flags)
        assert c == co and e == eo and c > 1000
        assert np.array_equal(g, go)
        assert np.array_equal(h, ho)
        assert np.array_equal(h, h.T)


@pytest.mark.parametrize("n", [1, 31, 127, 128, 1000, 65536 + 77])
def test_hgec_ragged_sizes(n):
    """point counts around the reference's launch-shape edges (N<128, N%32 != 0, N>65536), both modes."""
    reg, oa, pts, res = build_scene(rings=16, az=128)
    rng = np.random.default_rng(n)
    q = pts[rng.integers(0, pts.shape[0], n)]
    reg.reg_.prepare_registration(q)
    T = S.perturbation(12, -7, 3, 0.8)
    for flags in (0, 1):
        reg.reg_.flags = flags
        h, g, e, c = reg.reg_.perform_registration(reg.tsdf().device_map(), T, res)
        ho, go, eo, co = O.reg_iterate(oa, T, q, res, flags)
        assert (e, c) == (eo, co)
        assert np.array_equal(g, go) and np.array_equal(h, ho)


@pytest.mark.parametrize("loop", ["resident", "launches"])
def test_register_cloud_matches_oracle_pose(loop):
    """register_cloud: same iteration count, per-iteration sums and final pose (1e-4 m / 1e-4 rad),
    with the loop as one resident launch (grid barrier) and as one launch per iteration."""
    import warpsense_amd as W
    reg, oa, pts, res = build_scene(scans=2)
    reg.reg_.set_loop(W.WS_REG_LOOP_RESIDENT if loop == "resident" else W.WS_REG_LOOP_LAUNCHES)
    Tp = S.perturbation(60, 40, 0, 3.0)
    q = S.transform_points_mm(pts, Tp)
    T_gpu = reg.register_cloud(q, np.eye(4, dtype=np.float32))
    T_cpu, it_cpu, _ = O.register_cloud(oa, q, np.eye(4), 200, 0.1, 0.03, res)
    dt, ang = pose_error(T_gpu, T_cpu)
    assert reg.last_iterations == it_cpu, (reg.last_iterations, it_cpu)
    assert dt < 1e-4 and ang < 1e-4, (dt, ang)
    # and it actually registers: closer to the inverse perturbation than the start
    inv = np.linalg.inv(Tp.astype(np.float64))
    assert pose_error(T_gpu, inv)[0] < pose_error(np.eye(4), inv)[0]


def test_register_cloud_loop_modes_identical():
    """resident loop == per-iteration launches, bit for bit (pose and iteration count), incl. max_iterations cut-offs."""
    import warpsense_amd as W
    reg, oa, pts, res = build_scene(rings=32, az=256)
    q = S.transform_points_mm(pts, S.perturbation(-45, 25, 5, -2.0))
    reg.reg_.prepare_registration(q)
    for max_it in (0, 1, 2, 7, 200):
        out = []
        for mode in (W.WS_REG_LOOP_RESIDENT, W.WS_REG_LOOP_LAUNCHES):
            reg.reg_.set_loop(mode)
            T, it = reg.reg_.register_cloud(reg.tsdf().device_map(), np.eye(4, dtype=np.float32), max_it, 0.1, 0.03, res)
            out.append((T, it))
        assert out[0][1] == out[1][1] and out[0][1] <= max_it
        assert np.array_equal(out[0][0], out[1][0])
        if max_it == 0:
            assert np.array_equal(out[0][0], np.eye(4, dtype=np.float32))


def test_resident_loop_times_out_and_the_registration_is_repeated():
    """One workgroup's contribution never arrives (as if another kernel kept it off the chip): the exchange gives up after
    5 ms, ws_register_cloud repeats the registration with one launch per iteration -- same pose, same iteration count --
    and the next resident registration is clean again (the accumulators of the aborted launch are cleared by its successor)."""
    import ctypes as C
    reg, oa, pts, res = build_scene(rings=32, az=256)
    q = S.transform_points_mm(pts, S.perturbation(-45, 25, 5, -2.0))
    r = reg.reg_
    r.prepare_registration(q)
    args = (reg.tsdf().device_map(), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, res)
    T0, it0 = r.register_cloud(*args)
    before = C.c_int32(0)
    assert r._L.ws_debug_reg_stall(r.handle, 1, C.byref(before)) == 0
    T1, it1 = r.register_cloud(*args)
    after = C.c_int32(0)
    assert r._L.ws_debug_reg_stall(r.handle, 0, C.byref(after)) == 0
    assert after.value == before.value + 1
    assert it1 == it0 and np.array_equal(T1, T0)
    for _ in range(3):  # both accumulator sets come round again
        T2, it2 = r.register_cloud(*args)
        assert it2 == it0 and np.array_equal(T2, T0)
    done = C.c_int32(0)
    r._L.ws_debug_reg_stall(r.handle, 0, C.byref(done))
    assert done.value == after.value


def test_resident_loop_with_a_co_tenant_kernel():
    """Another stream keeps the chip busy (a train of large GEMMs) while registrations run: the resident loop needs all of its
    256 workgroups on the chip at once, so it either gets them or gives up within 5 ms and ws_register_cloud repeats the
    registration with one launch per iteration -- in both cases the pose and the iteration count are the ones of a quiet GPU."""
    import ctypes as C
    import torch
    reg, oa, pts, res = build_scene(rings=32, az=256)
    q = S.transform_points_mm(pts, S.perturbation(-45, 25, 5, -2.0))
    r = reg.reg_
    r.prepare_registration(q)
    args = (reg.tsdf().device_map(), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, res)
    T0, it0 = r.register_cloud(*args)
    before = C.c_int32(0)
    r._L.ws_debug_reg_stall(r.handle, 0, C.byref(before))
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(60):
            a = (a @ a).clamp_(-1, 1)
    got = [r.register_cloud(*args) for _ in range(6)]
    side.synchronize()
    after = C.c_int32(0)
    r._L.ws_debug_reg_stall(r.handle, 0, C.byref(after))
    print("registrations repeated with one launch per iteration under the co-tenant:", after.value - before.value, "of", len(got))
    for T, it in got:
        assert it == it0 and np.array_equal(T, T0)
    T1, it1 = r.register_cloud(*args)  # and alone again
    assert it1 == it0 and np.array_equal(T1, T0)


def test_register_cloud_empty_overlap():
    """c == 0 (cloud far outside the map): the loop stops instead of producing NaN."""
    reg, oa, pts, res = build_scene(rings=8, az=64)
    q = pts + np.array([10_000_000, 0, 0], dtype=np.int32)
    T = reg.register_cloud(q[:100], np.eye(4, dtype=np.float32))
    assert np.all(np.isfinite(T))
    assert np.allclose(T, np.eye(4))


@pytest.mark.parametrize("n", [0, 5, 200_000])
def test_register_cloud_point_count_extremes(n):
    """no points at all, fewer points than lanes in one wave, and more than two passes of the resident grid
    (131 072 lanes x 2 points): same iteration count and pose as the oracle in both loop modes."""
    import warpsense_amd as W
    reg, oa, pts, res = build_scene(rings=32, az=256)
    rng = np.random.default_rng(7)
    q = S.transform_points_mm(pts, S.perturbation(25, -18, 6, 1.2))
    q = q[rng.integers(0, len(q), n)] if n else q[:0]
    T_cpu, it_cpu, _ = O.register_cloud(oa, q, np.eye(4), 200, 0.1, 0.03, res)
    for mode in (W.WS_REG_LOOP_RESIDENT, W.WS_REG_LOOP_LAUNCHES):
        reg.reg_.set_loop(mode)
        T = reg.register_cloud(q, np.eye(4, dtype=np.float32))
        assert reg.last_iterations == it_cpu, (mode, reg.last_iterations, it_cpu)
        assert np.all(np.isfinite(T))
        dt, ang = pose_error(T, T_cpu)
        assert dt < 1e-4 and ang < 1e-4, (mode, dt, ang)


def test_wave_solver_is_bit_identical_to_the_oracle_lu():
    """solve6_wave (one matrix element per lane) == wso_solve6 (serial LU, the same operations in the same order), bit
    for bit: normal equations as the registration produces them, matrices that need every pivot choice, badly scaled
    and singular ones."""
    import ctypes as C
    import warpsense_amd as W
    from warpsense_amd import _lib
    rng = np.random.default_rng(17)
    mats, rhs = [], []
    for _ in range(400):  # J^T J of integer Jacobians + damping, like gn_update builds it
        J = np.concatenate([rng.integers(-2_000_000, 2_000_000, size=(50, 3)), rng.integers(-600, 600, size=(50, 3))], axis=1).astype(np.float64)
        mats.append(J.T @ J + np.eye(6) * float(rng.integers(0, 5000)))
        rhs.append(J.T @ rng.integers(-1000, 1000, size=50).astype(np.float64))
    for _ in range(400):  # general matrices: pivoting in every column
        mats.append(rng.normal(size=(6, 6)) * 10.0 ** rng.integers(-8, 9, size=(6, 1)))
        rhs.append(rng.normal(size=6))
    for _ in range(100):  # exact ties in the pivot column and zeros on the diagonal
        mats.append(rng.integers(-2, 3, size=(6, 6)).astype(np.float64))
        rhs.append(rng.integers(-3, 4, size=6).astype(np.float64))
    mats.append(np.zeros((6, 6)))
    rhs.append(np.ones(6))
    A = np.ascontiguousarray(np.stack(mats), dtype=np.float64)
    b = np.ascontiguousarray(np.stack(rhs), dtype=np.float64)
    n = len(A)
    x = np.zeros((n, 6), dtype=np.float64)
    st = np.zeros(n, dtype=np.int32)
    ctx = W.Context.default()
    _lib.check(ctx._L.ws_debug_solve6(ctx.handle, A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), n, x.ctypes.data_as(C.c_void_p),
                                     st.ctypes.data_as(C.c_void_p)), "ws_debug_solve6")
    L = O.lib()
    singular = 0
    for i in range(n):
        xo = np.zeros(6, dtype=np.float64)
        rc = L.wso_solve6(A[i].ctypes.data_as(C.c_void_p), b[i].ctypes.data_as(C.c_void_p), xo.ctypes.data_as(C.c_void_p))
        assert rc == st[i], i
        if rc != 0:
            singular += 1
            continue
        assert np.array_equal(x[i].view(np.uint64), xo.view(np.uint64)), (i, x[i], xo)
    assert 1 <= singular < 60


def test_sharded_driver_on_one_rank_equals_the_resident_loop():
    """warpsense_amd.dist.sharded_register_cloud (accumulate -> [all-reduce] -> solve per iteration, the multi-GPU
    driver) on a single rank without a process group: identical pose and iteration count as ws_register_cloud."""
    import warpsense_amd as W
    from warpsense_amd.dist import HipGnBackend, sharded_register_cloud
    reg, oa, pts, res = build_scene(rings=32, az=256)
    q = S.transform_points_mm(pts, S.perturbation(30, -22, 7, 1.4))
    reg.reg_.prepare_registration(q)
    T1, it1 = reg.reg_.register_cloud(reg.tsdf().device_map(), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, res)
    backend = HipGnBackend(reg.reg_, reg.tsdf(), res)
    T2, it2 = sharded_register_cloud(backend, len(q), np.eye(4, dtype=np.float32), 200, 0.1, 0.03)
    assert it1 == it2 and np.array_equal(T1, T2)


@pytest.mark.parametrize("n,res", [(4096, 2000), (131072, 2000), (1000, 50), (200_000, 2000)])
def test_loop_sums_on_a_map_of_arbitrary_entries(n, res):
    """The resident loop's sums on inputs that use the whole width of the arithmetic.  The map holds RANDOM raw entries: values
    over all of int16 (incl. -32768 / 32767), half of them same-sign extremes next to each other (gradients of +-16383), weights
    of any sign, a fifth unobserved.  The points lie up to 120 m out and the pretransform carries a 54 m translation, so the
    integer transform wraps for many of them (cu_transform_point, cuda/util.h:11-22: int32) and the cross products q x gradient
    cover all of int32 -- the test asserts values beyond +-0x7f7f7f80, where a signed-limb split without a bias would fail.
    Clouds of at most 131 072 points take the matrix-core route (J as byte limbs, v_mfma_i32_32x32x32_i8), the larger one the
    v_mad_i64_i32 route: h, g, e, c of the last iteration, the pose and the iteration count must be the oracle's, bit for bit,
    in both loop modes."""
    import warpsense_amd as W
    rng = np.random.default_rng(1000 + n)
    size = (65, 65, 65) if res >= 900 else (129, 129, 65)
    n_vox = size[0] * size[1] * size[2]
    value = rng.integers(-32768, 32768, n_vox).astype(np.int64)
    ext = np.where(rng.random(n_vox) < 0.5, 32767, 1)
    value = np.where(rng.random(n_vox) < 0.5, np.where((np.arange(n_vox) // (size[1] * size[2])) % 2 == 0, ext, -ext - 1), value)
    weight = rng.integers(-32768, 32768, n_vox).astype(np.int64)
    weight[rng.random(n_vox) < 0.2] = 0
    raw = ((value & 0xffff) | ((weight & 0xffff) << 16)).astype(np.uint32)
    om = O.OracleMap(size, 0, 0, data=raw.copy())
    view = W.DeviceMap(om.size.copy(), om.offset.copy(), raw.copy(), om.pos.copy())
    tsdf = W.TSDFCuda(view, 1000, 640, res)
    reach = 120_000 if res >= 900 else 3_000
    q = rng.integers(-reach, reach + 1, (n, 3)).astype(np.int32)
    T_in = S.perturbation(40_000, -30_000, 20_000, 3.0) if res >= 900 else S.perturbation(20, -15, 10, 3.0)
    if n == 131072:
        J = np.asarray(O.calc_jacobis(om, T_in, q, res)[0]).astype(np.int64)
        assert (J[:, :3] > 0x7F7F7F80).any() and (J[:, :3] < -0x7F7F7F80).any()
    rc = W.RegistrationCuda(None, tsdf.ctx)
    rc.prepare_registration(q)
    for max_it in (1, 2, 6):
        T_o, it_o, trace = O.register_cloud(om, q, T_in, max_it, 0.1, 0.03, res, trace_cap=8)
        assert it_o >= 1
        ho = trace[it_o - 1][:36].reshape(6, 6).T
        go, eo, co = trace[it_o - 1][36:42], int(trace[it_o - 1][42]), int(trace[it_o - 1][43])
        assert co > n // 4 and np.abs(ho).max() > 2 ** 50  # observed voxels under most points; sums far beyond 32 bits
        for mode in (W.WS_REG_LOOP_RESIDENT, W.WS_REG_LOOP_LAUNCHES):
            rc.set_loop(mode)
            T, it = rc.register_cloud(tsdf.device_map(), T_in, max_it, 0.1, 0.03, res)
            h, g, e, c = rc.last_sums()
            assert it == it_o, (mode, max_it, it, it_o)
            assert (e, c) == (eo, co), (mode, max_it, e, eo, c, co)
            assert np.array_equal(g, go), (mode, max_it)
            assert np.array_equal(h, ho), (mode, max_it)
            assert np.array_equal(T, T_o.astype(np.float32)), (mode, max_it, np.abs(T - T_o).max())
    # the same inputs through the resident server behind perform_registration (matrix cores up to 131 072 points, v_mad_i64_i32 beyond),
    # twice: the second request answers from the voxels the first one cached
    _server(rc, enable=1, idle_us=200000)
    ho, go, eo, co = O.reg_iterate(om, T_in, q, res, rc.flags)
    for _ in range(2):
        h, g, e, c = rc.perform_registration(tsdf.device_map(), T_in, res)
        assert (e, c) == (eo, co) and np.array_equal(g, go) and np.array_equal(h, ho)
    T2 = (S.perturbation(300, 200, -100, 0.7) @ np.asarray(T_in, dtype=np.float64)).astype(np.float32)
    ho, go, eo, co = O.reg_iterate(om, T2, q, res, rc.flags)
    h, g, e, c = rc.perform_registration(tsdf.device_map(), T2, res)
    assert (e, c) == (eo, co) and np.array_equal(g, go) and np.array_equal(h, ho)
    rc.close()


def _server(reg_cuda, enable=-1, idle_us=0):
    import ctypes as C
    n = C.c_int32(0)
    rc = reg_cuda._L.ws_debug_reg_server(reg_cuda.handle, int(enable), int(idle_us), C.byref(n))
    assert rc == 0
    return n.value


def test_perform_registration_through_the_resident_server():
    """the reference's caller is unchanged -- one perform_registration per Gauss-Newton iteration, the solve on the
    host in between (tsdf_registration.cpp:55-92) -- and a launch per call cost 19.8 us.  ws_reg_iterate now talks to a kernel
    that stays on the GPU across calls (reg_server_kernel): the pose goes out and the 44 sums come back through host-mapped
    memory.  Bit-exact against the oracle for a sequence of poses, ONE launch for the whole sequence while the calls keep coming,
    the same numbers with the server switched off, and a new server after the old one has gone idle."""
    import time
    reg, oa, pts, res = build_scene()
    q = S.transform_points_mm(pts, S.perturbation(40, -30, 10, 2.0))
    rc = reg.reg_
    rc.prepare_registration(q)
    poses = [np.eye(4, dtype=np.float32)] + [S.perturbation(3.0 * k - 20, 28 - k, -9 + 0.5 * k, 0.15 * k - 1.7) for k in range(24)]
    want = [O.reg_iterate(oa, T, q, res, rc.flags) for T in poses]
    # (objects of earlier tests that die in the middle of the sequence would enqueue work on the shared context -- freeing a
    # map synchronises its stream -- and that asks the server to leave, as it must: keep the collector out of the sequence)
    import gc
    gc.collect()
    gc.disable()
    try:
        base = _server(rc, enable=1, idle_us=200000)  # (a Python caller is slow: keep the server through the gaps)
        got = [rc.perform_registration(reg.tsdf().device_map(), T, res) for T in poses]
        assert _server(rc) - base == 1, "one server for the whole sequence"
    finally:
        gc.enable()
    for (h, g, e, c), (ho, go, eo, co) in zip(got, want):
        assert c == co and e == eo and c > 1000 and np.array_equal(g, go) and np.array_equal(h, ho)
    # the same calls with one launch each
    _server(rc, enable=0)
    for T, (ho, go, eo, co) in zip(poses[:5], want):
        h, g, e, c = rc.perform_registration(reg.tsdf().device_map(), T, res)
        assert c == co and e == eo and np.array_equal(g, go) and np.array_equal(h, ho)
    # a short idle time: the server has left between two calls, the next call starts another
    base = _server(rc, enable=1, idle_us=20)
    for T, (ho, go, eo, co) in zip(poses[:6], want):
        h, g, e, c = rc.perform_registration(reg.tsdf().device_map(), T, res)
        time.sleep(0.002)
        assert c == co and e == eo and np.array_equal(g, go) and np.array_equal(h, ho)
    assert _server(rc) - base >= 5


def test_the_resident_server_sees_what_happens_between_two_calls():
    """A living server answers from the cloud it holds in registers and from the map as the stream ordered it: a new cloud
    (prepare_registration), an update of the map, a download -- everything that enqueues work on the context's stream asks the
    server to leave, and the next perform_registration starts a new one BEHIND that work.  Every answer is the oracle's for the
    state the caller has built up."""
    import torch
    reg, oa, pts, res = build_scene(rings=32, az=256)
    on = O.OracleMap(oa.size, 1000, 0)
    rc = reg.reg_
    _server(rc, enable=1, idle_us=200000)
    T = S.perturbation(12, -7, 3, 0.8)
    q1 = S.transform_points_mm(pts, S.perturbation(40, -30, 10, 2.0))
    q2 = np.ascontiguousarray(S.transform_points_mm(pts, S.perturbation(-25, 14, -6, -1.1))[::-1][: len(pts) - 77])
    pts2 = S.os1_128_scan(rings=32, azimuths=256, half_extents_mm=(2400.0, 2500.0, 900.0), seed=77)

    def check(q):
        h, g, e, c = rc.perform_registration(reg.tsdf().device_map(), T, res)
        ho, go, eo, co = O.reg_iterate(oa, T, q, res, rc.flags)
        assert c == co and e == eo and np.array_equal(g, go) and np.array_equal(h, ho)

    rc.prepare_registration(q1)
    check(q1)
    check(q1)
    rc.prepare_registration(torch.from_numpy(q2).cuda())  # another cloud
    check(q2)
    reg.update_tsdf(torch.from_numpy(pts2).cuda(), pose=np.eye(4, dtype=np.float32))  # the map changes
    O.update_tsdf(oa, on, pts2, (0, 0, 0), (0, 0, 32768), 1000, 640, res)
    check(q2)
    check(q2)
    reg.tsdf().ctx.sync()
    check(q2)


def test_a_reader_thread_beside_the_resident_server():
    """the reference's shift thread reads the map (avg_map().to_host, tsdf_mapping.cpp:115-117) while the registration thread is
    in its loop: a download from a second host thread asks the server to leave in the middle of a sequence of requests; no
    request is lost or answered twice, every answer is exact"""
    import threading
    import warpsense_amd as W
    reg, oa, pts, res = build_scene(rings=32, az=256)
    rc = reg.reg_
    _server(rc, enable=1, idle_us=300)
    q = S.transform_points_mm(pts, S.perturbation(40, -30, 10, 2.0))
    rc.prepare_registration(q)
    poses = [S.perturbation(2.0 * k - 20, 18 - k, -9 + 0.5 * k, 0.1 * k - 1.2) for k in range(40)]
    want = [O.reg_iterate(oa, T, q, res, rc.flags) for T in poses]
    stop = threading.Event()
    errs = []
    lm = W.LocalMap(*oa.size, 1000, 0)
    n_reads = [0]

    def reader():
        try:
            host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
            while not stop.is_set():
                reg.tsdf().avg_map().to_host(host)
                n_reads[0] += 1
                assert np.array_equal(host.data_, oa.data)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=reader)
    th.start()
    try:
        for rep in range(3):
            for T, (ho, go, eo, co) in zip(poses, want):
                h, g, e, c = rc.perform_registration(reg.tsdf().device_map(), T, res)
                assert c == co and e == eo and np.array_equal(g, go) and np.array_equal(h, ho)
    finally:
        stop.set()
        th.join()
    assert not errs, errs
    assert n_reads[0] > 0


def test_registrations_come_and_go_beside_a_reader_thread():
    """every entry point that takes a map walks the context's list of registrations (a resident server must leave before other work
    is enqueued): the list is guarded -- another thread may create and destroy registrations of the same context meanwhile"""
    import threading
    import warpsense_amd as W
    reg, oa, pts, res = build_scene(rings=32, az=256)
    ctx = reg.tsdf().ctx
    lm = W.LocalMap(*oa.size, 1000, 0)
    host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
    stop = threading.Event()
    errs = []

    def churn():
        try:
            while not stop.is_set():
                rcs = [W.RegistrationCuda(None, ctx) for _ in range(4)]
                for r in rcs:
                    r.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=churn)
    th.start()
    try:
        for _ in range(60):
            reg.tsdf().avg_map().to_host(host)
            assert np.array_equal(host.data_, oa.data)
    finally:
        stop.set()
        th.join()
    assert not errs, errs


def test_the_resident_server_behind_a_long_upload_of_another_thread():
    """another thread keeps uploading a 270 MB map (tens of milliseconds each, far beyond the time a caller spins for the server's
    answer): the server is asked to leave, a new one waits in the stream behind the copy -- the call drains the stream and takes the
    answer then; every answer is exact, none is an error.  (The uploads write the map's own content back: the answers do not change.)"""
    import threading
    import warpsense_amd as W
    reg, oa, pts, res = build_scene(size=(512, 512, 256), rings=32, az=256)
    rc = reg.reg_
    _server(rc, enable=1, idle_us=300)
    q = S.transform_points_mm(pts, S.perturbation(40, -30, 10, 2.0))
    rc.prepare_registration(q)
    poses = [S.perturbation(2.0 * k - 20, 18 - k, -9 + 0.5 * k, 0.1 * k - 1.2) for k in range(12)]
    want = [O.reg_iterate(oa, T, q, res, rc.flags) for T in poses]
    host = W.DeviceMap(oa.size.copy(), oa.offset.copy(), oa.data.copy(), oa.pos.copy())
    stop = threading.Event()
    errs = []
    n_up = [0]

    def uploader():
        try:
            while not stop.is_set():
                reg.tsdf().avg_map().to_device(host)
                n_up[0] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=uploader)
    th.start()
    try:
        for rep in range(3):
            for T, (ho, go, eo, co) in zip(poses, want):
                h, g, e, c = rc.perform_registration(reg.tsdf().device_map(), T, res)
                assert c == co and e == eo and np.array_equal(g, go) and np.array_equal(h, ho)
    finally:
        stop.set()
        th.join()
    assert not errs, errs
    assert n_up[0] > 0
