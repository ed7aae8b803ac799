"""Point-sharded Point-to-TSDF registration across GPUs.

The map is replicated; rank r owns the contiguous point range [r*N/G, (r+1)*N/G) of the cloud.  Every
Gauss-Newton iteration each rank accumulates its 44 int64 partial sums (h 6x6 column-major, g[6], e, c),
the partials are summed with ONE all-reduce (352 B; RCCL over xGMI on the GPUs, gloo in the CPU tests) and
every rank runs the identical 6x6 solve.  Integer sums are exact, so the result is bit-identical for any
number of ranks.

The per-rank compute is behind a small backend interface; the product backend (HipGnBackend) drives the
C ABI (ws_reg_begin / ws_reg_iterate_shard_dev -- or ws_reg_accumulate_dev + ws_reg_solve_dev -- / ws_reg_poll).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check


GRAPH_CACHE_MAX = 8  # captured batches kept by sharded_register_cloud's `graphs` dict
PEER_TIMEOUTS_BEFORE_GIVING_UP = 2  # consecutive exchange time-outs after which a group stops using the device-side route


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous index range of `rank`: [rank*n//world, (rank+1)*n//world)."""
    lo = (rank * n) // world
    hi = ((rank + 1) * n) // world
    return lo, hi - lo


class HipGnBackend:
    """Device-side Gauss-Newton building blocks of one rank (all stream-ordered, no host sync but poll)."""

    def __init__(self, reg, tsdf, map_resolution: int, flags: int = 0):
        import torch
        self.reg, self.tsdf, self.res, self.flags = reg, tsdf, int(map_resolution), int(flags)
        self._L = reg._L
        # one stream for the library, torch and the hand-off to the collective: the 44 words are written by the library's
        # kernels and read by the all-reduce (RCCL kernel or host copy) with nothing but stream order in between
        reg.ctx.use_torch_stream()
        self.sums = torch.zeros(44, dtype=torch.int64, device="cuda")
        self._pending = False  # iterate(): sums of an iteration whose update has not been applied yet
        self.peers = None      # (rank, world) once connect_peers() / connect_local() has mapped the mailboxes

    def begin(self, T_in, max_iterations, it_weight_gradient, epsilon):
        T = np.ascontiguousarray(np.asarray(T_in, dtype=np.float32).reshape(4, 4).T).reshape(16)
        check(self._L.ws_reg_begin(self.reg.handle, T.ctypes.data_as(C.c_void_p), int(max_iterations),
                                   C.c_float(it_weight_gradient), C.c_float(epsilon)), "ws_reg_begin")
        self._pending = False

    def accumulate(self, first: int, count: int):
        check(self._L.ws_reg_accumulate_dev(self.reg.handle, self.tsdf.device_map(), self.res, self.flags, int(first),
                                            int(count), C.c_void_p(self.sums.data_ptr())), "ws_reg_accumulate_dev")
        return self.sums

    def solve(self, sums):
        check(self._L.ws_reg_solve_dev(self.reg.handle, C.c_void_p(sums.data_ptr())), "ws_reg_solve_dev")
        self._pending = False

    def iterate(self, first: int, count: int):
        """One launch per iteration: the update from the (all-reduced) sums of the previous iterate(), if there was one
        since begin() / solve(), then the accumulation of the shard into the same 44 words (ws_reg_iterate_shard_dev).
        The caller all-reduces the returned tensor and, after the last iteration of a batch, calls solve() on it."""
        check(self._L.ws_reg_iterate_shard_dev(self.reg.handle, self.tsdf.device_map(), self.res, self.flags, int(first), int(count),
                                               C.c_void_p(self.sums.data_ptr()), 1 if self._pending else 0), "ws_reg_iterate_shard_dev")
        self._pending = True
        return self.sums

    def binding(self) -> tuple:
        """Everything a captured kernel launch bakes in by value: the map window (size, pos, offset travel as kernel
        arguments), the voxel array and the registration's point buffer and count.  A HIP graph captured under one
        binding must not be replayed under another (stale window / freed buffer)."""
        par = np.zeros((3, 3), dtype=np.int32)
        check(self._L.ws_map_get_params(self.tsdf.device_map(), 0, par[0].ctypes.data_as(C.c_void_p), par[1].ctypes.data_as(C.c_void_p),
                                        par[2].ctypes.data_as(C.c_void_p)), "ws_map_get_params")
        n = C.c_size_t(0)
        pts = self._L.ws_reg_points_dev(self.reg.handle, C.byref(n))
        return (tuple(int(v) for v in par.reshape(-1)), int(self._L.ws_map_device_data(self.tsdf.device_map(), 0) or 0), int(pts or 0), int(n.value),
                self.res, self.flags)

    # ---- the loop without the host in it: the ranks' sums meet in mailboxes in each other's HBM (ws_reg_peer_*) ----------
    def connect_peers(self, group=None, blocks: int = 0):
        """Once per process group: export this rank's mailbox as an IPC handle, all-gather the handles, map the peers'.
        `blocks`: workgroups of the resident loop on this rank (0 = 256, one per CU; ranks that share one GPU must split
        the chip: 256 // ranks-per-GPU).  Afterwards sharded_register_cloud runs ws_register_cloud_peers."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world > 8:
            raise ValueError("the device-side exchange connects at most 8 ranks (one node); larger groups use the RCCL route")
        handle = (C.c_ubyte * 64)()
        check(self._L.ws_reg_peer_mailbox(self.reg.handle, handle), "ws_reg_peer_mailbox")
        handles = [bytes(handle)]
        if world > 1:
            handles = [None] * world
            dist.all_gather_object(handles, bytes(handle), group=group)
        blob = b"".join(handles)
        err = None
        rc = self._L.ws_reg_peer_connect(self.reg.handle, rank, world, blob, int(blocks))
        if rc != 0:
            msg = self._L.ws_last_error()
            err = f"ws_reg_peer_connect failed with status {rc}: {msg.decode(errors='replace') if msg else ''}"
        if world > 1:
            # all or nobody: a rank that could not map a peer's mailbox must not leave the others waiting in the resident loop
            import torch
            flag = torch.tensor([0 if err else 1], dtype=torch.int32)
            if dist.get_backend(group) != "gloo":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0 and err is None:
                err = "a peer rank could not map the mailboxes"
        if err is not None:
            self._L.ws_reg_peer_disconnect(self.reg.handle)
            self.peers = None
            from ._lib import WsError
            raise WsError(err)
        if world > 1:
            dist.barrier(group=group)  # every mailbox is mapped (and zero) before anybody's first launch
        self.peers = (rank, world)

    def connect_local(self, backends, rank: int, blocks: int = 0):
        """the same for several HipGnBackend objects of ONE process (tests: several contexts on one GPU), without IPC"""
        arr = (C.c_void_p * len(backends))(*[b.reg.handle for b in backends])
        check(self._L.ws_reg_peer_connect_local(self.reg.handle, rank, len(backends), arr, int(blocks)), "ws_reg_peer_connect_local")
        self.peers = (rank, len(backends))

    def reset_peers(self):
        check(self._L.ws_reg_peer_reset(self.reg.handle), "ws_reg_peer_reset")

    def drop_peers(self):
        """leave the device-side route for good (sharded_register_cloud after repeated time-outs): unmap the peers' mailboxes"""
        check(self._L.ws_reg_peer_disconnect(self.reg.handle), "ws_reg_peer_disconnect")
        self.peers = None

    def register_peers(self, first: int, count: int, T_in, max_iterations, it_weight_gradient, epsilon):
        """ws_register_cloud_peers: the whole Gauss-Newton loop in one launch per rank.  Returns (T 4x4, iterations), or None
        if the exchange with the peers timed out (WS_ERR_TIMEOUT)."""
        T = np.ascontiguousarray(np.asarray(T_in, dtype=np.float32).reshape(4, 4).T).reshape(16)
        out = np.empty(16, dtype=np.float32)
        it = C.c_int32(0)
        rc = self._L.ws_register_cloud_peers(self.reg.handle, self.tsdf.device_map(), int(first), int(count), T.ctypes.data_as(C.c_void_p),
                                             int(max_iterations), C.c_float(it_weight_gradient), C.c_float(epsilon), self.res, self.flags,
                                             out.ctypes.data_as(C.c_void_p), C.byref(it))
        if rc == -6:  # WS_ERR_TIMEOUT
            return None
        check(rc, "ws_register_cloud_peers")
        return out.reshape(4, 4).T.copy(), int(it.value)

    def poll(self):
        fin, it = C.c_int32(0), C.c_int32(0)
        out = np.zeros(16, dtype=np.float32)
        check(self._L.ws_reg_poll(self.reg.handle, C.byref(fin), C.byref(it), out.ctypes.data_as(C.c_void_p)), "ws_reg_poll")
        return bool(fin.value), int(it.value), out.reshape(4, 4).T.copy()


def all_reduce_sums(sums, group=None):
    """Sum the 44 int64 words over the ranks of `group`, in place.  RCCL ("nccl") reduces the device tensor over xGMI;
    a gloo group (the CPU tests, and two processes that share ONE GPU in the GPU tests) gets a host copy -- 352 bytes."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return sums
    if sums.is_cuda and dist.get_backend(group) == "gloo":
        host = sums.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        sums.copy_(host)
        return sums
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


class _GraphBatch:
    """`batch` iterations of [accumulate -> all-reduce -> solve] captured once as a HIP graph and replayed: the host
    launches one graph per batch instead of four operations per iteration (kernels that find the loop finished return
    at once, so replaying past convergence is harmless).  Falls back to eager launches if capture is not possible."""

    def __init__(self, backend, first, count, group, batch):
        import torch
        import torch.distributed as dist
        self.graph = None
        self.batch = batch
        self._run = lambda: self._eager(backend, first, count, group, batch)
        if not hasattr(backend, "reg"):
            return  # CPU test backend
        ctx = backend.reg.ctx
        home = torch.cuda.current_stream().cuda_stream
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # one eager batch on the side stream (RCCL must be warm before a capture)
                ctx.set_stream(side.cuda_stream)
                self._eager(backend, first, count, group, 1)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                self._eager(backend, first, count, group, batch)
            ctx.set_stream(home)
            self.graph = g
            self._run = g.replay
        except Exception as exc:  # capture unsupported (old RCCL, a backend that synchronises): stay eager
            import os
            import sys
            if os.environ.get("WS_DIST_DEBUG"):
                print(f"[warpsense_amd.dist] graph capture failed, staying eager: {exc!r}", file=sys.stderr)
            ctx.set_stream(home)
            self.graph = None
        # every rank must take the same route (a graph replay and eager launches issue the same collectives, but
        # agreeing keeps the ranks in lockstep if capture only failed somewhere)
        if dist.is_initialized():
            ok = torch.tensor([1 if self.graph is not None else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0 and self.graph is not None:
                self.graph = None
                self._run = lambda: self._eager(backend, first, count, group, batch)

    @staticmethod
    def _eager(backend, first, count, group, n):
        import torch.distributed as dist
        if hasattr(backend, "iterate") and n > 0:
            # one kernel + one all-reduce per iteration (the update of iteration i rides in the launch of iteration i + 1)
            for _ in range(n):
                sums = backend.iterate(first, count)
                all_reduce_sums(sums, group)
            backend.solve(sums)  # the last update of the batch, so that poll() sees the state after n iterations
            return
        for _ in range(n):
            sums = backend.accumulate(first, count)
            all_reduce_sums(sums, group)
            backend.solve(sums)

    def run(self):
        self._run()

    def validate(self, backend, first, count, group, T_in, it_weight_gradient, epsilon) -> bool:
        """A captured batch must do exactly what the same launches do on a stream (integer sums: bit for bit).  ROCm 7.0
        replays consecutive kernel nodes without the cache maintenance a stream gives consecutive kernels; the kernels
        here exchange their results at agent scope because of that, and this check catches a runtime where even that is
        not enough: on a mismatch the batch falls back to eager launches (all ranks together)."""
        if self.graph is None:
            return True
        import torch
        import torch.distributed as dist
        ok = True
        for _ in range(2):
            backend.begin(T_in, self.batch, it_weight_gradient, epsilon)
            self._eager(backend, first, count, group, self.batch)
            a = backend.poll()
            backend.begin(T_in, self.batch, it_weight_gradient, epsilon)
            self.graph.replay()
            b = backend.poll()
            ok = ok and a[1] == b[1] and np.array_equal(a[2], b[2])
        if dist.is_initialized():
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            ok = bool(int(flag.item()))
        if not ok:
            import sys
            print("[warpsense_amd.dist] a replayed HIP graph of Gauss-Newton iterations differs from the same launches on a "
                  "stream: staying with eager launches", file=sys.stderr)
            self.graph = None
            batch = self.batch
            self._run = lambda: self._eager(backend, first, count, group, batch)
        return ok


def sharded_register_cloud(backend, n_points: int, T_in, max_iterations: int, it_weight_gradient: float, epsilon: float,
                           group=None, batch: int = 16, graphs: dict | None = None):
    """cuda::TSDFRegistration::register_cloud (tsdf_registration.cpp:28-96) with the points sharded over the
    ranks of `group`.  Returns (total_transform 4x4, iterations).  Every rank returns the same values.

    `graphs`: a dict the caller keeps between calls; with it the per-batch work is captured once as a HIP graph and
    replayed.  The key holds everything the captured launches bake in (shard, batch size, map window, voxel and point
    buffers, point count — HipGnBackend.binding()), so a map shift or a cloud of another size captures anew instead of
    replaying against a stale window; the dict keeps the GRAPH_CACHE_MAX most recent graphs."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = shard_range(n_points, rank, world)
    if getattr(backend, "peers", None) is not None and backend.peers[1] == world:
        # the resident loop on every rank, sums exchanged device to device: one launch per registration
        # (an error other than the time-out -- e.g. a sticky map error surfacing here -- must not leave the other ranks
        # blocked in the agreement below: vote "not ok", take part in the collective, raise afterwards)
        failure = None
        try:
            res = backend.register_peers(first, count, T_in, max_iterations, it_weight_gradient, epsilon)
        except Exception as exc:  # noqa: BLE001 - re-raised below, after the collective
            res, failure = None, exc
        ok = res is not None
        # the ranks agree on ONE of three verdicts before anyone takes another route: 2 = the exchange worked, 1 = it timed out
        # (every rank sees that), 0 = a rank failed with an ERROR (with a two-valued vote the other ranks went on
        # into the all-reduce route and blocked in collectives the failed rank, which re-raised, never joined)
        verdict = 2 if ok else (0 if failure is not None else 1)
        if world > 1:
            import torch
            flag = torch.tensor([verdict], dtype=torch.int32)
            if dist.get_backend(group) != "gloo":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            verdict = int(flag.item())
            ok = verdict == 2
        if verdict == 0:
            # every rank leaves the call here, together: the failing rank with its own exception, the others with this one
            backend.reset_peers()
            if world > 1:
                dist.barrier(group=group)
            if failure is not None:
                raise failure
            raise RuntimeError("sharded_register_cloud: a peer rank failed in the device-side registration (its own exception says why); "
                               "no rank took the all-reduce route")
        if ok:
            backend.peer_timeouts = 0
            backend.last_route = "device_mailboxes"
            return res
        import sys
        # every rank saw the same verdict above, so every rank counts alike and they leave the route together
        backend.peer_timeouts = getattr(backend, "peer_timeouts", 0) + 1
        give_up = backend.peer_timeouts >= PEER_TIMEOUTS_BEFORE_GIVING_UP
        print("[warpsense_amd.dist] the device-side exchange timed out; this registration runs through the all-reduce route"
              + (" and so do all later ones (time-out %d in a row)" % backend.peer_timeouts if give_up else ""), file=sys.stderr)
        backend.peer_timeouts_total = getattr(backend, "peer_timeouts_total", 0) + 1
        if give_up:
            backend.drop_peers()
        else:
            backend.reset_peers()
        if world > 1:
            dist.barrier(group=group)
    backend.last_route = "all_reduce"
    runner = None
    if graphs is not None:
        key = (first, count, batch) + (backend.binding() if hasattr(backend, "binding") else ())
        runner = graphs.pop(key, None)
        if runner is None:
            backend.begin(T_in, 1, it_weight_gradient, epsilon)  # a throw-away state for the warm-up / capture launches
            runner = _GraphBatch(backend, first, count, group, batch)
            if hasattr(backend, "reg"):
                runner.validate(backend, first, count, group, T_in, it_weight_gradient, epsilon)
        graphs[key] = runner  # most recently used last
        while len(graphs) > GRAPH_CACHE_MAX:
            graphs.pop(next(iter(graphs)))
    backend.begin(T_in, max_iterations, it_weight_gradient, epsilon)
    done, finished, iterations, T = 0, False, 0, np.asarray(T_in, dtype=np.float32)
    while not finished and done < max_iterations:
        if runner is not None:
            runner.run()  # whole batches: iterations beyond max_iterations / convergence are no-ops on the device
            done += batch
        else:
            todo = min(batch, max_iterations - done)
            _GraphBatch._eager(backend, first, count, group, todo)
            done += todo
        finished, iterations, T = backend.poll()
    return T, iterations
