"""Point-sharded Point-to-TSDF registration across GPUs.

The map is replicated; rank r owns the contiguous point range [r*N/G, (r+1)*N/G) of the cloud.  Every
Gauss-Newton iteration each rank accumulates its 44 int64 partial sums (h 6x6 column-major, g[6], e, c),
the partials are summed with ONE all-reduce (352 B; RCCL over xGMI on the GPUs, gloo in the CPU tests) and
every rank runs the identical 6x6 solve.  Integer sums are exact, so the result is bit-identical for any
number of ranks.

The per-rank compute is behind a small backend interface; the product backend (HipGnBackend) drives the
C ABI (ws_reg_begin / ws_reg_accumulate_dev / ws_reg_solve_dev / ws_reg_poll).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check


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
        self.sums = torch.zeros(44, dtype=torch.int64, device="cuda")

    def begin(self, T_in, max_iterations, it_weight_gradient, epsilon):
        T = np.ascontiguousarray(np.asarray(T_in, dtype=np.float32).reshape(4, 4).T).reshape(16)
        check(self._L.ws_reg_begin(self.reg.handle, T.ctypes.data_as(C.c_void_p), int(max_iterations),
                                   C.c_float(it_weight_gradient), C.c_float(epsilon)), "ws_reg_begin")

    def accumulate(self, first: int, count: int):
        check(self._L.ws_reg_accumulate_dev(self.reg.handle, self.tsdf.device_map(), self.res, self.flags, int(first),
                                            int(count), C.c_void_p(self.sums.data_ptr())), "ws_reg_accumulate_dev")
        return self.sums

    def solve(self, sums):
        check(self._L.ws_reg_solve_dev(self.reg.handle, C.c_void_p(sums.data_ptr())), "ws_reg_solve_dev")

    def poll(self):
        fin, it = C.c_int32(0), C.c_int32(0)
        out = np.zeros(16, dtype=np.float32)
        check(self._L.ws_reg_poll(self.reg.handle, C.byref(fin), C.byref(it), out.ctypes.data_as(C.c_void_p)), "ws_reg_poll")
        return bool(fin.value), int(it.value), out.reshape(4, 4).T.copy()


def sharded_register_cloud(backend, n_points: int, T_in, max_iterations: int, it_weight_gradient: float, epsilon: float,
                           group=None, batch: int = 16):
    """cuda::TSDFRegistration::register_cloud (tsdf_registration.cpp:28-96) with the points sharded over the
    ranks of `group`.  Returns (total_transform 4x4, iterations).  Every rank returns the same values."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = shard_range(n_points, rank, world)
    backend.begin(T_in, max_iterations, it_weight_gradient, epsilon)
    done, finished, iterations, T = 0, False, 0, np.asarray(T_in, dtype=np.float32)
    while not finished and done < max_iterations:
        todo = min(batch, max_iterations - done)
        for _ in range(todo):
            sums = backend.accumulate(first, count)
            if dist.is_initialized():
                dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
            backend.solve(sums)
        done += todo
        finished, iterations, T = backend.poll()
    return T, iterations
