"""CPU tests of the boundary: the C-ABI library loads and exports every declared symbol (no compute without
a GPU), the host-side helpers match the oracle, and the multi-rank registration driver is exact (gloo)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from warpsense_amd import _lib
    header = open(os.path.join(ROOT, "include", "warpsense_hip.h")).read()
    declared = set(re.findall(r"\b(ws_[a-z0-9_]+)\s*\(", header))
    declared -= {"ws_status"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.load()  # loads without a GPU: no device call at load time
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.ws_version() >= 1


def test_no_cpu_fallback_in_product_package():
    """nothing under warpsense_amd/ may import or link the oracle."""
    pkg = os.path.join(ROOT, "warpsense_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in text and "libws_oracle" not in text and "ws_oracle.h" not in text, f


def test_convert_pose_matches_oracle():
    import warpsense_amd as W
    rng = np.random.default_rng(1)
    params = W.Params(W.MapParams(resolution=50))
    tm = W.TSDFMapping.__new__(W.TSDFMapping)
    tm.params_ = params
    for _ in range(50):
        a, b, c = rng.uniform(-0.6, 0.6, 3)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = (Rz @ Ry @ Rx).astype(np.float32)
        T[:3, 3] = rng.uniform(-9000, 9000, 3).astype(np.float32)
        pos, up = tm.convert_pose_to_gpu(T)
        opos, oup = O.convert_pose(T, 50)
        assert np.array_equal(pos, opos) and np.array_equal(up, oup)


def test_map_params_scaling_rules():
    """include/params/map_params.h:100-114: tau = max_distance*1000, max_weight *= 64, size = metres*1000/res."""
    import warpsense_amd as W
    p = W.MapParams(resolution=64, max_distance=1.0, max_weight=10, size=(40.0, 40.0, 25.0))
    assert (p.tau, p.max_weight, p.size) == (1000, 640, (625, 625, 390))


def test_shard_ranges_partition_the_cloud():
    from warpsense_amd.dist import shard_range
    for n in (0, 1, 7, 131072, 100003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b


# ------------------------------------------------------------------ multi-rank driver over gloo (world size 2)
class OracleGnBackend:
    """Test double for HipGnBackend: per-rank partial sums and the solve come from the CPU oracle, so the
    sharding + all-reduce + identical-solve logic of warpsense_amd.dist runs without a GPU."""

    class State(C.Structure):
        _fields_ = [("T", C.c_float * 16), ("center", C.c_int32 * 3), ("alpha", C.c_float), ("prev", C.c_float * 4),
                    ("it_weight_gradient", C.c_float), ("epsilon", C.c_float), ("max_iterations", C.c_int32),
                    ("iterations", C.c_int32), ("finished", C.c_int32)]

    def __init__(self, omap, points, res):
        import torch
        self.m, self.pts, self.res = omap, np.ascontiguousarray(points, dtype=np.int32), res
        self.st = self.State()
        self.sums = torch.zeros(44, dtype=torch.int64)

    def begin(self, T_in, max_iterations, it_weight_gradient, epsilon):
        T = O.colmajor(T_in)
        O.lib().wso_gn_begin(C.byref(self.st), O._p(T), int(max_iterations), C.c_float(it_weight_gradient), C.c_float(epsilon))

    def accumulate(self, first, count):
        T = np.ctypeslib.as_array(self.st.T).reshape(4, 4).T
        if self.st.finished or self.st.iterations >= self.st.max_iterations:
            return self.sums
        h, g, e, c = O.reg_iterate(self.m, T, self.pts[first:first + count], self.res)
        flat = np.concatenate([h.T.reshape(-1), g, [e, c]]).astype(np.int64)
        self.sums.copy_(__import__("torch").from_numpy(flat))
        return self.sums

    def solve(self, sums):
        s = np.ascontiguousarray(sums.numpy(), dtype=np.int64)
        O.lib().wso_gn_update(C.byref(self.st), O._p(s))

    def poll(self):
        fin = bool(self.st.finished or self.st.iterations >= self.st.max_iterations)
        return fin, int(self.st.iterations), np.ctypeslib.as_array(self.st.T).reshape(4, 4).T.copy()


def _gloo_worker(rank, world, port, q, drop=0):
    import torch.distributed as dist
    from warpsense_amd import synthetic as S
    from warpsense_amd.dist import sharded_register_cloud
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tau, res, mw, size = 1000, 50, 640, (96, 96, 48)
        pts = S.os1_128_scan(rings=32, azimuths=128, half_extents_mm=(2000.0, 1700.0, 800.0), seed=5)
        avg = O.OracleMap(size, tau, 0)
        new = avg.copy()
        O.update_tsdf(avg, new, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
        cloud = S.transform_points_mm(pts, S.perturbation(25, -15, 5, 1.2))
        if drop:
            cloud = np.ascontiguousarray(cloud[:-drop])  # ragged: the point count is not a multiple of the world size
        backend = OracleGnBackend(avg, cloud, res)
        T, it = sharded_register_cloud(backend, cloud.shape[0], np.eye(4, dtype=np.float32), 60, 0.1, 0.03, batch=7)
        if rank == 0:
            assert cloud.shape[0] % world != 0 or not drop
            T_ref, it_ref, _ = O.register_cloud(avg, cloud, np.eye(4), 60, 0.1, 0.03, res)
            q.put((it, it_ref, float(np.abs(T - T_ref).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,drop", [(2, 0), (2, 1), (3, 0), (3, 2)])
def test_sharded_registration_is_exact_over_gloo(world, drop):
    """2 and 3 ranks, points sharded by index (also when the count is not a multiple of the world size: shard_range
    gives the ranks unequal, contiguous ranges), all-reduce of the 44 int64 sums each iteration: bit-identical to 1 rank."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q, drop)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    it, it_ref, err = q.get(timeout=10)
    assert it == it_ref and it > 3
    assert err == 0.0


class FlakyPeerBackend(OracleGnBackend):
    """OracleGnBackend that also offers the device-side route -- and whose exchange never completes on rank 1 (a node whose
    mapped mailboxes do not carry the atomics): what sharded_register_cloud does about it is host logic, testable here."""

    def __init__(self, omap, points, res, rank, world):
        super().__init__(omap, points, res)
        self.rank, self.peers, self.calls, self.resets, self.drops = rank, (rank, world), 0, 0, 0

    def register_peers(self, first, count, T_in, max_iterations, it_weight_gradient, epsilon):
        self.calls += 1
        return None if self.rank == 1 else (np.full((4, 4), 7.0, dtype=np.float32), 1)  # rank 0 "finished" with something else

    def reset_peers(self):
        self.resets += 1

    def drop_peers(self):
        self.drops += 1
        self.peers = None


def _flaky_worker(rank, world, port, q):
    import torch.distributed as dist
    from warpsense_amd import synthetic as S
    from warpsense_amd.dist import PEER_TIMEOUTS_BEFORE_GIVING_UP, sharded_register_cloud
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tau, res, mw, size = 1000, 50, 640, (96, 96, 48)
        pts = S.os1_128_scan(rings=32, azimuths=128, half_extents_mm=(2000.0, 1700.0, 800.0), seed=5)
        avg = O.OracleMap(size, tau, 0)
        new = avg.copy()
        O.update_tsdf(avg, new, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
        cloud = S.transform_points_mm(pts, S.perturbation(25, -15, 5, 1.2))
        backend = FlakyPeerBackend(avg, cloud, res, rank, world)
        T_ref, it_ref, _ = O.register_cloud(avg, cloud, np.eye(4), 60, 0.1, 0.03, res)
        exact = []
        for _ in range(PEER_TIMEOUTS_BEFORE_GIVING_UP + 2):
            T, it = sharded_register_cloud(backend, cloud.shape[0], np.eye(4, dtype=np.float32), 60, 0.1, 0.03, batch=7)
            exact.append(it == it_ref and bool(np.array_equal(T, T_ref.astype(np.float32))))
        q.put((rank, exact, backend.calls, backend.resets, backend.drops, backend.peers, PEER_TIMEOUTS_BEFORE_GIVING_UP))
    finally:
        dist.destroy_process_group()


def test_a_group_whose_device_side_exchange_keeps_timing_out_leaves_that_route_together():
    """rank 1's exchange times out every time, rank 0's does not: every registration still ends exact on BOTH ranks (the ranks
    agree on the verdict before anyone returns, so rank 0's own result is discarded), and after
    PEER_TIMEOUTS_BEFORE_GIVING_UP time-outs in a row both ranks drop the route for good -- no 0.25 s stall per scan."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flaky_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank, exact, calls, resets, drops, peers, limit in sorted(q.get(timeout=10) for _ in range(2)):
        assert all(exact) and len(exact) == limit + 2, (rank, exact)
        assert calls == limit and resets == limit - 1 and drops == 1 and peers is None, (rank, calls, resets, drops, peers)


class FailingPeerBackend(FlakyPeerBackend):
    """the device-side route RAISES on rank 1 (a sticky map error surfacing there), works on rank 0"""

    def register_peers(self, first, count, T_in, max_iterations, it_weight_gradient, epsilon):
        self.calls += 1
        if self.rank == 1:
            raise ValueError("rank 1: the map is not exact")
        return np.full((4, 4), 7.0, dtype=np.float32), 1


def _failing_worker(rank, world, port, q):
    import torch.distributed as dist
    from warpsense_amd import synthetic as S
    from warpsense_amd.dist import sharded_register_cloud
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tau, res, mw, size = 1000, 50, 640, (64, 64, 32)
        pts = S.os1_128_scan(rings=8, azimuths=64, half_extents_mm=(1200.0, 1000.0, 500.0), seed=5)
        avg = O.OracleMap(size, tau, 0)
        new = avg.copy()
        O.update_tsdf(avg, new, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
        backend = FailingPeerBackend(avg, pts, res, rank, world)
        try:
            sharded_register_cloud(backend, pts.shape[0], np.eye(4, dtype=np.float32), 20, 0.1, 0.03, batch=7)
            q.put((rank, "returned", backend.resets))
        except ValueError as exc:
            q.put((rank, "own:" + str(exc), backend.resets))
        except RuntimeError as exc:
            q.put((rank, "peer:" + str(exc)[:40], backend.resets))
    finally:
        dist.destroy_process_group()


def test_an_error_on_one_rank_ends_the_call_on_every_rank():
    """when the device-side route raises on one rank, the others must not walk into the all-reduce route alone (they
    blocked there until the backend's time-out).  The vote has three values: every rank leaves the call together -- the failing
    rank with its own exception, the others with one that says a peer failed."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = dict((r, (what, resets)) for r, what, resets in (q.get(timeout=10) for _ in range(2)))
    assert got[1][0].startswith("own:rank 1") and got[0][0].startswith("peer:"), got
    assert got[0][1] == 1 and got[1][1] == 1


def test_bench_gpus_flag_starts_the_ranks_or_checks_the_launcher(monkeypatch):
    """bench.py --gpus N was parsed and never read.  Without a launcher N > 1 now re-executes under
    torch.distributed.run with N ranks (rendezvous on 127.0.0.1, the caller's flags passed on); under a launcher the world size
    must be the --gpus asked for."""
    import argparse
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("ws_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []
    monkeypatch.setattr(os, "execv", lambda exe, argv: calls.append((exe, list(argv))))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    bench.self_launch(argparse.Namespace(gpus=4))
    assert len(calls) == 1
    argv = calls[0][1]
    assert argv[1:3] == ["-m", "torch.distributed.run"] and argv[argv.index("--nproc-per-node") + 1] == "4"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert os.path.basename(argv[-7]) == "bench.py"
    # one GPU: nothing to launch
    calls.clear()
    bench.self_launch(argparse.Namespace(gpus=1))
    assert not calls
    # under a launcher: the world must match
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.self_launch(argparse.Namespace(gpus=4))
    assert not calls
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit):
        bench.self_launch(argparse.Namespace(gpus=4))


def test_the_servers_mail_protocol_on_the_host_side():
    """the resident server behind ws_reg_iterate answers in seven 64-byte lines of seven words + a tag (request number, hash of the
    words) that it writes without waiting: the host must not take a stale answer, one that has not arrived completely, or a torn
    line -- exercised on ordinary memory, no GPU (ws_debug_reg_mail_selftest returns a bit per failed case)"""
    from warpsense_amd import _lib
    assert _lib.load().ws_debug_reg_mail_selftest() == 0
