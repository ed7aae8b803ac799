"""Two host threads on one map, in the lock discipline of the reference's callers (tsdf_mapping.cpp:62-75,114-124, tsdf_registration.cpp:54):

  thread A (the scan callback)   register the scan (shared lock), then update_tsdf with it (unique lock), scan after scan
  thread B (the map-shift thread) avg_map().to_host (shared lock) in a loop, every few rounds to_device + update_params of what it has
                                  just read (unique lock): the content of the map does not change by that

The same sequence of scans is run once alone and once beside thread B, for each way thread A can register (the resident loop behind
register_cloud; one perform_registration per Gauss-Newton iteration through the resident server, the solve on the host in between;
one launch per perform_registration): poses, iteration counts and the final map must be the same bit for bit.

    python tools/stress_threads.py [--scans 12] [--size 256]
"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class SharedMutex:
    """std::shared_mutex, enough of it; a waiting writer keeps new readers out (thread B re-takes the shared lock at once: without
    that thread A would never get its turn)"""

    def __init__(self):
        self._c = threading.Condition()
        self._readers = 0
        self._writer = False
        self._waiting = 0

    def lock_shared(self):
        with self._c:
            while self._writer or self._waiting:
                self._c.wait()
            self._readers += 1

    def unlock_shared(self):
        with self._c:
            self._readers -= 1
            self._c.notify_all()

    def lock(self):
        with self._c:
            self._waiting += 1
            while self._writer or self._readers:
                self._c.wait()
            self._waiting -= 1
            self._writer = True

    def unlock(self):
        with self._c:
            self._writer = False
            self._c.notify_all()


def run(mode, scans, size, with_b, writeback=True):
    import torch
    import oracle_lib as O
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    from test_abi_and_host import OracleGnBackend
    tau, res, mw = 1000, 50, 640
    shape = (size, size, size // 2)
    params = W.Params(W.MapParams(resolution=res, max_distance=tau / 1000.0, max_weight=mw // 64, size=tuple(s * res / 1000.0 for s in shape)))
    lm = W.LocalMap(shape[0], shape[1], shape[2], tau, 0)
    reg = W.TSDFRegistration(params, lm)
    rc = reg.reg_
    n = C.c_int32(0)
    rc._L.ws_debug_reg_server(rc.handle, 1 if mode == "server" else 0, 300, C.byref(n))
    he = tuple(0.4 * s * res for s in shape)
    mu = SharedMutex()
    stop = threading.Event()
    errs = []
    counts = {"reads": 0, "writes": 0}

    def thread_b():
        try:
            host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
            k = 0
            while not stop.is_set():
                mu.lock_shared()
                try:
                    reg.tsdf().avg_map().to_host(host)
                finally:
                    mu.unlock_shared()
                counts["reads"] += 1
                k += 1
                if writeback and k % 3 == 0:
                    mu.lock()
                    try:
                        # (what it has read is still the map: nobody else writes under the unique lock)
                        reg.tsdf().avg_map().to_host(host)
                        reg.tsdf().avg_map().to_device(host)
                        reg.tsdf().new_map().update_params(host)
                    finally:
                        mu.unlock()
                    counts["writes"] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=thread_b)
    if with_b:
        th.start()
    poses, its = [], []
    try:
        pose = np.eye(4, dtype=np.float32)
        for k in range(scans):
            pts = S.os1_128_scan(rings=64, azimuths=512, half_extents_mm=he, seed=100 + k)
            q = S.transform_points_mm(pts, S.perturbation(20 + 3 * k, -15 + 2 * k, 4, 1.0 + 0.2 * k))
            if k > 0:
                mu.lock_shared()
                try:
                    if mode == "loop":
                        T = reg.register_cloud(q, np.eye(4, dtype=np.float32))
                        it = reg.last_iterations
                    else:
                        # the reference's own loop (tsdf_registration.cpp:55-92) around perform_registration, the oracle's host update in between
                        st = OracleGnBackend.State()
                        rp = reg.params_.registration
                        O.lib().wso_gn_begin(C.byref(st), O._p(O.colmajor(np.eye(4, dtype=np.float32))), int(rp.max_iterations), C.c_float(rp.it_weight_gradient),
                                             C.c_float(rp.epsilon))
                        rc.prepare_registration(q)
                        while not (st.finished or st.iterations >= st.max_iterations):
                            Tk = np.ctypeslib.as_array(st.T).reshape(4, 4).T.copy()
                            h, g, e, c = rc.perform_registration(reg.tsdf().device_map(), Tk, res)
                            sums = np.concatenate([h.T.reshape(-1), g, [e, c]]).astype(np.int64)
                            O.lib().wso_gn_update(C.byref(st), O._p(np.ascontiguousarray(sums)))
                        T = np.ctypeslib.as_array(st.T).reshape(4, 4).T.copy()
                        it = int(st.iterations)
                finally:
                    mu.unlock_shared()
                poses.append(np.asarray(T, dtype=np.float32).copy())
                its.append(it)
            mu.lock()
            try:
                reg.update_tsdf(torch.from_numpy(pts).cuda(), pose=pose)
            finally:
                mu.unlock()
    finally:
        stop.set()
        if with_b:
            th.join()
    if errs:
        raise errs[0]
    host = W.DeviceMap(lm.size.copy(), lm.offset.copy(), np.empty_like(lm.data), lm.pos.copy())
    reg.tsdf().avg_map().to_host(host)
    return poses, its, host.data_.copy(), counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=12)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--modes", default="loop,server,launches")
    ap.add_argument("--no-writeback", action="store_true", help="thread B only reads")
    ap.add_argument("--limit", type=int, default=240, help="seconds per mode before the stacks are dumped and the run ends")
    args = ap.parse_args()
    bad = 0
    import faulthandler
    for mode in args.modes.split(","):
        faulthandler.dump_traceback_later(args.limit, exit=True)  # (a hang is a finding: say where)
        t0 = time.time()
        p0, i0, m0, _ = run(mode, args.scans, args.size, False)
        p1, i1, m1, counts = run(mode, args.scans, args.size, True, not args.no_writeback)
        same = i0 == i1 and all(np.array_equal(a, b) for a, b in zip(p0, p1)) and np.array_equal(m0, m1)
        bad += 0 if same else 1
        print(f"{mode:9s}: {args.scans} scans, iterations {i0[:6]}..., thread B read the map {counts['reads']} times and wrote it back {counts['writes']} times "
              f"-> {'identical' if same else 'DIFFERENT'}  ({time.time() - t0:.1f} s)", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
