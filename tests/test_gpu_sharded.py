"""SURVEY §8e on real kernels: the point-sharded registration's HIP entry points on PARTIAL ranges (first > 0, count < N).

One GPU is enough to run what G ranks would run: every "rank" is its own ws_reg handle (own Gauss-Newton state, own 44-word
buffer) holding the whole prepared cloud and accumulating only its range [r*N/G, (r+1)*N/G); the test plays the all-reduce
(adds the G partials on the host, writes the total back into every rank's buffer) and compares, iteration by iteration,
with the oracle's loop (registration.cu:347-368 x tsdf_registration.cpp:55-92).  A second test runs the real driver
(warpsense_amd.dist.sharded_register_cloud, the `iterate` route) in two processes that share cuda:0 and all-reduce over gloo.
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import oracle_lib as O
from warpsense_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _scene(rings=32, az=256):
    import test_gpu_registration as R
    reg, oa, pts, res = R.build_scene(rings=rings, az=az)
    q = S.transform_points_mm(pts, S.perturbation(30, -22, 7, 1.4))
    return reg, oa, q, res


class _OracleLoop:
    """the oracle's Gauss-Newton loop, one step at a time (wso_gn_begin / wso_reg_iterate / wso_gn_update)"""

    def __init__(self, omap, points, res, T_in, max_iterations, it_weight_gradient, epsilon):
        from test_abi_and_host import OracleGnBackend
        self.b = OracleGnBackend(omap, points, res)
        self.b.begin(T_in, max_iterations, it_weight_gradient, epsilon)
        self.n = points.shape[0]

    def finished(self):
        return self.b.poll()[0]

    def sums(self):
        return self.b.accumulate(0, self.n).numpy().copy()

    def update(self, sums):
        import torch
        self.b.solve(torch.from_numpy(np.ascontiguousarray(sums, dtype=np.int64)))

    def result(self):
        return self.b.poll()


@pytest.mark.parametrize("route", ["iterate", "accumulate"])
@pytest.mark.parametrize("world,drop", [(2, 1), (3, 0), (3, 1), (8, 5)])
def test_hip_shard_ranges_sum_to_the_whole(world, drop, route):
    """G in {2, 3, 8} ranges (ragged: N is not a multiple of G) through ws_reg_iterate_shard_dev (one launch per iteration,
    the update of iteration i at the head of launch i + 1) and through ws_reg_accumulate_dev + ws_reg_solve_dev: the sum of
    the ranks' 44 words == the oracle's h, g, e, c in EVERY iteration, and every rank ends with the oracle's iteration count
    and a bit-identical pose."""
    import torch
    import warpsense_amd as W
    from warpsense_amd.dist import HipGnBackend, shard_range
    reg, oa, q, res = _scene()
    if drop:
        q = np.ascontiguousarray(q[:-drop])
    n = q.shape[0]
    assert drop == 0 or n % world != 0
    T0 = np.eye(4, dtype=np.float32)
    args = (200, 0.1, 0.03)
    ranks = []
    for r in range(world):
        rc = W.RegistrationCuda(ctx=reg.tsdf().ctx)
        rc.prepare_registration(q)  # every rank holds the whole cloud and works on its range
        ranks.append(HipGnBackend(rc, reg.tsdf(), res))
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[-1][0] > 0 and all(c < n for _, c in spans)
    for b in ranks:
        b.begin(T0, *args)
    oracle = _OracleLoop(oa, q, res, T0, *args)
    its = 0
    while not oracle.finished():
        want = oracle.sums()
        parts = []
        for b, (first, count) in zip(ranks, spans):
            s = b.iterate(first, count) if route == "iterate" else b.accumulate(first, count)
            parts.append(s.cpu().numpy().copy())
        total = np.sum(parts, axis=0, dtype=np.int64)
        assert np.array_equal(total, want), (its, np.nonzero(total != want)[0])
        assert total[43] > 1000 and all(p[43] > 0 for p in parts)  # every range contributes correspondences
        tt = torch.from_numpy(total)
        for b in ranks:
            b.sums.copy_(tt)  # the all-reduce
            if route == "accumulate":
                b.solve(b.sums)
        oracle.update(total)
        its += 1
    if route == "iterate":
        for b in ranks:
            b.solve(b.sums)  # the last update of the loop (what sharded_register_cloud does at the end of a batch)
    fin_o, it_o, T_o = oracle.result()
    assert fin_o and it_o == its > 5
    for b in ranks:
        fin, it, T = b.poll()
        assert fin and it == it_o
        assert np.array_equal(T, T_o), np.abs(T - T_o).max()
    # and the same loop on one rank / one launch for everything: the resident kernel
    reg.reg_.prepare_registration(q)
    T1, it1 = reg.reg_.register_cloud(reg.tsdf().device_map(), T0, *args, res)
    assert it1 == it_o and np.array_equal(T1, T_o)
    for b in ranks:
        b.reg.close()


def test_shard_launch_after_convergence_changes_nothing():
    """launches enqueued past convergence (a replayed batch does that) neither move the state nor touch the sums"""
    import warpsense_amd as W
    from warpsense_amd.dist import HipGnBackend, sharded_register_cloud
    reg, oa, q, res = _scene()
    reg.reg_.prepare_registration(q)
    b = HipGnBackend(reg.reg_, reg.tsdf(), res)
    T, it = sharded_register_cloud(b, len(q), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, batch=16)
    sums = b.sums.cpu().numpy().copy()
    for _ in range(3):
        b.iterate(0, len(q))
    fin, it2, T2 = b.poll()
    assert fin and it2 == it and np.array_equal(T2, T)
    assert np.array_equal(b.sums.cpu().numpy(), sums)
    T_o, it_o, _ = O.register_cloud(oa, q, np.eye(4), 200, 0.1, 0.03, res)
    assert it == it_o and np.array_equal(T, T_o.astype(np.float32))


def test_sharded_driver_with_hip_graphs_is_exact_or_falls_back():
    """the opt-in HIP-graph route of the driver: a captured batch of shard launches replays from whatever state the single
    state buffer holds (no host-side parity baked in); it validates itself against the same launches on a stream and falls
    back to them if the runtime replays differently -- either way the result is the oracle's"""
    from warpsense_amd.dist import HipGnBackend, sharded_register_cloud
    reg, oa, q, res = _scene()
    reg.reg_.prepare_registration(q)
    b = HipGnBackend(reg.reg_, reg.tsdf(), res)
    graphs = {}
    T_o, it_o, _ = O.register_cloud(oa, q, np.eye(4), 200, 0.1, 0.03, res)
    for _ in range(3):
        T, it = sharded_register_cloud(b, len(q), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, batch=16, graphs=graphs)
        assert it == it_o and np.array_equal(T, T_o.astype(np.float32))
    runner = next(iter(graphs.values()))
    print("HIP graph route in use:", runner.graph is not None)


def _two_rank_worker(rank, world, port, q_out, drop):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)  # both ranks on the one GPU of the box
        from warpsense_amd.dist import HipGnBackend, shard_range, sharded_register_cloud
        reg, oa, q, res = _scene()
        if drop:
            q = np.ascontiguousarray(q[:-drop])
        reg.reg_.prepare_registration(q)
        backend = HipGnBackend(reg.reg_, reg.tsdf(), res)
        first, count = shard_range(len(q), rank, world)
        T, it = sharded_register_cloud(backend, len(q), np.eye(4, dtype=np.float32), 200, 0.1, 0.03, batch=7)
        T_o, it_o, _ = O.register_cloud(oa, q, np.eye(4), 200, 0.1, 0.03, res)
        q_out.put((rank, first, count, it, it_o, bool(np.array_equal(T, T_o.astype(np.float32))), float(np.abs(T - T_o).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("drop", [0, 1])
def test_sharded_register_cloud_two_processes_on_one_gpu(drop):
    """world = 2: two processes, both on cuda:0, each with its own context, map and ws_reg; sharded_register_cloud's
    one-launch-per-iteration route (HipGnBackend.iterate -> ws_reg_iterate_shard_dev with first > 0 on rank 1) with the
    44 words all-reduced over gloo.  Both ranks must end with the oracle's iteration count and pose, bit for bit."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q_out, drop)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    got = sorted(q_out.get(timeout=10) for _ in range(2))
    assert got[0][1] == 0 and got[1][1] == got[0][2] > 0  # rank 1 starts behind rank 0's range
    for rank, first, count, it, it_o, same, err in got:
        assert it == it_o > 5, (rank, it, it_o)
        assert same, (rank, err)


# ---------------------------------------------------------------------------------------------------------------------
# the sharded loop WITHOUT the host in it: resident loop per rank, sums exchanged through mailboxes in HBM (ws_reg_peer_*)
# ---------------------------------------------------------------------------------------------------------------------
def _peer_ranks_in_one_process(reg, q, res, world, blocks):
    """`world` ranks as ws_reg handles on their own contexts (own streams) of one process, connected without IPC"""
    import warpsense_amd as W
    from warpsense_amd.dist import HipGnBackend
    ranks = []
    for r in range(world):
        ctx = W.Context(-1)  # own stream: the ranks' resident kernels must be on the chip together
        rc = W.RegistrationCuda(ctx=ctx)
        rc.prepare_registration(q)
        b = HipGnBackend.__new__(HipGnBackend)  # (no torch-stream hand-off: this route has no host step per iteration)
        b.reg, b.tsdf, b.res, b.flags, b._L, b.peers, b._pending = rc, reg.tsdf(), res, 0, rc._L, None, False
        ranks.append(b)
    for r, b in enumerate(ranks):
        b.connect_local(ranks, r, blocks)
    return ranks


def _run_ranks(ranks, n, args):
    import threading
    from warpsense_amd.dist import shard_range
    world = len(ranks)
    out = [None] * world

    def work(r):
        first, count = shard_range(n, r, world)
        out[r] = ranks[r].register_peers(first, count, np.eye(4, dtype=np.float32), *args)  # ctypes drops the GIL: the calls overlap

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    return out


# The ranks of these tests are ws_reg handles on separate streams of ONE process, and their resident kernels must be on the chip
# together.  A process maps its streams onto 4 hardware queues by default, and streams that share a queue run one after the
# other: inside the full test session (dozens of streams by then) two ranks can land on one queue and wait for each other
# until the poll limit.  So the checks below (inner_*: not collected by the session) run one by one in pytest processes of
# their own with 16 hardware queues, started by test_peer_mailbox_loops_in_a_process_of_their_own.


@pytest.mark.parametrize("case", ["peer_mailbox_loop_equals_the_oracle[2-128-1]", "peer_mailbox_loop_equals_the_oracle[3-80-2]",
                                  "peer_mailbox_loop_equals_the_oracle[4-64-3]", "peer_mailbox_loop_equals_the_oracle[8-32-5]",
                                  "peer_mailbox_loop_equals_the_oracle[1-256-0]", "peer_loop_times_out_when_a_rank_is_missing_and_recovers"])
def test_peer_mailbox_loops_in_a_process_of_their_own(case):
    import subprocess
    import sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__) + "::inner_" + case, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-o", "python_functions=inner_*"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "1 passed" in out, out[-3000:]


@pytest.mark.parametrize("world,blocks,drop", [(2, 128, 1), (3, 80, 2), (4, 64, 3), (8, 32, 5), (1, 256, 0)])
def inner_peer_mailbox_loop_equals_the_oracle(world, blocks, drop):
    """2 / 3 / 4 / 8 ranks share cuda:0 (at most 256 / world resident workgroups each, so that all are on the chip at once): every rank
    runs the resident Gauss-Newton loop on its ragged shard, the ranks' 44 sums meet in counted mailboxes, nobody talks to the
    host in between.  Every rank must return the oracle's iteration count and pose, bit for bit, registration after
    registration (the mailbox words are never reset: each launch starts counting where the last one stopped)."""
    reg, oa, q, res = _scene()
    if drop:
        q = np.ascontiguousarray(q[:-drop])
    n = q.shape[0]
    args = (200, 0.1, 0.03)
    ranks = _peer_ranks_in_one_process(reg, q, res, world, blocks)
    T_o, it_o, _ = O.register_cloud(oa, q, np.eye(4), *args, res)
    for rep in range(3):
        out = _run_ranks(ranks, n, args)
        for r, got in enumerate(out):
            assert got is not None, f"rank {r} timed out (repetition {rep})"
            T, it = got
            assert it == it_o > 5, (rep, r, it, it_o)
            assert np.array_equal(T, T_o), (rep, r, np.abs(T - T_o).max())
    # cut-offs: every rank stops at the same iteration
    for max_it in (0, 1, 7):
        out = _run_ranks(ranks, n, (max_it, 0.1, 0.03))
        T_c, it_c, _ = O.register_cloud(oa, q, np.eye(4), max_it, 0.1, 0.03, res)
        for got in out:
            assert got is not None and got[1] == it_c <= max_it and np.array_equal(got[0], T_c)
    for b in ranks:
        b.reg.close()
        b.reg.ctx.close()  # and its stream: the hardware queues of the process are few


def inner_peer_loop_times_out_when_a_rank_is_missing_and_recovers():
    """rank 1 never launches: rank 0 gives up after the 0.25 s poll limit (WS_ERR_TIMEOUT -> None) instead of hanging; after
    ws_reg_peer_reset on both, the pair registers exactly again"""
    reg, oa, q, res = _scene()
    n = q.shape[0]
    args = (200, 0.1, 0.03)
    ranks = _peer_ranks_in_one_process(reg, q, res, 2, 128)
    from warpsense_amd.dist import shard_range
    first, count = shard_range(n, 0, 2)
    assert ranks[0].register_peers(first, count, np.eye(4, dtype=np.float32), *args) is None
    for b in ranks:
        b.reset_peers()
    T_o, it_o, _ = O.register_cloud(oa, q, np.eye(4), *args, res)
    for got in _run_ranks(ranks, n, args):
        assert got is not None and got[1] == it_o and np.array_equal(got[0], T_o)
    for b in ranks:
        b.reg.close()
        b.reg.ctx.close()


def _peer_worker(rank, world, port, q_out, drop):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from warpsense_amd.dist import HipGnBackend, sharded_register_cloud
        reg, oa, q, res = _scene()
        if drop:
            q = np.ascontiguousarray(q[:-drop])
        reg.reg_.prepare_registration(q)
        backend = HipGnBackend(reg.reg_, reg.tsdf(), res)
        backend.connect_peers(blocks=256 // world)  # IPC handles all-gathered over gloo
        T_o, it_o, _ = O.register_cloud(oa, q, np.eye(4), 200, 0.1, 0.03, res)
        ok = True
        routes = []
        for _ in range(3):
            T, it = sharded_register_cloud(backend, len(q), np.eye(4, dtype=np.float32), 200, 0.1, 0.03)
            ok = ok and it == it_o and bool(np.array_equal(T, T_o))
            routes.append(getattr(backend, "last_route", None))
        q_out.put((rank, ok, it, it_o, routes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,drop", [(2, 1), (8, 3)])
def test_peer_mailbox_loop_processes_over_ipc(world, drop, monkeypatch):
    """the deployment shape on the one GPU of the box: `world` PROCESSES (2, and 8 = a whole node's ranks), mailboxes exported
    with hipIpcGetMemHandle, gathered over the process group and opened with hipIpcOpenMemHandle -- every rank maps all the
    others' -- and sharded_register_cloud on a ragged cloud: every rank ends with the oracle's iterations and pose, bit for bit.
    Two processes keep the device-side route; eight kernels of eight processes need not be on one GPU together, so there a
    registration may fall back to the all-reduce route (still exact) -- the connect itself must work."""
    import torch.multiprocessing as mp
    # ranks of different processes that SHARE a GPU start further apart than ranks with a GPU each (ws_reg_peer_connect reads this)
    monkeypatch.setenv("WS_REG_PEER_TIMEOUT_MS", "250")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, q_out, drop)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    got = sorted(q_out.get(timeout=10) for _ in range(world))
    for rank, ok, it, it_o, routes in got:
        assert ok and it == it_o > 5, (rank, ok, it, it_o, routes)
    if world == 2:
        assert all(r == "device_mailboxes" for _, _, _, _, routes in got for r in routes), got


def test_bench_dry_run_two_ranks_share_the_gpu():
    """bench.py's N > 1 path cannot rot while there is no multi-GPU box to run it on: two ranks launched the way
    the driver launches them, both on cuda:0 with a gloo group (WS_BENCH_SHARE_GPU=1 WS_BENCH_BACKEND=gloo) -- IPC mailboxes,
    device-side exchange, replica pass, reductions of the timings, and the keys the first real 8-GPU line will be read by."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WS_BENCH_SHARE_GPU="1", WS_BENCH_BACKEND="gloo", WS_REG_PEER_TIMEOUT_MS="250", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=root)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr.decode(errors="replace")[-3000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["dry_run_shared_gpu"] is True
    mg = out["multi_gpu"]
    assert mg["route"] in ("device_mailboxes", "all_reduce") and mg["iterations"] > 5 and mg["us_per_iteration"] > 0 and "exchange_timeouts" in mg
    assert out["replica_scans_per_s"] > 0


def test_bench_gpus_two_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher ran on one GPU and printed n_gpus: 1.  It now starts its
    N ranks itself (the same torch.distributed.run line); dry run with both ranks on cuda:0 over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WS_BENCH_SHARE_GPU="1", WS_BENCH_BACKEND="gloo", WS_REG_PEER_TIMEOUT_MS="250", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=root)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr.decode(errors="replace")[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["multi_gpu"]["iterations"] > 5


def test_peer_mailbox_loop_eight_ranks_on_one_gpu():
    """world = 8, the size of a node: eight ranks x 32 resident workgroups share cuda:0 (tools/peer_bench.py in a process of its
    own with 16 hardware queues, so that all eight kernels are on the chip together), the benchmark cloud of 131 072 points on the
    513^3 map: every rank ends with the one-rank resident loop's iteration count and pose, bit for bit"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "peer_bench.py"), "--ranks", "8", "--reps", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    assert lines, r.stderr.decode(errors="replace")[-2000:]
    out = json.loads(lines[-1])
    assert "error" not in out, out
    assert out["ranks"] == 8 and out["same_result_as_one_rank"] is True and out["iterations"] > 5
