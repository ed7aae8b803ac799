"""Randomised parity soak of register_cloud against the CPU oracle: random scenes and perturbations, the resident loop
must stop after the same number of Gauss-Newton iterations as the oracle and end within 1e-4 m / 1e-4 rad of its pose
(and equal the pose bit for bit in most cases -- counted).  The fixed cases live in tests/test_gpu_registration.py; this
is for changes to the first wave's arithmetic (solve, exponential map) and to the reduction.

    python tools/soak_reg.py [--cases 30] [--seed 1] [--server]

--server: the reference's OWN loop shape instead (tsdf_registration.cpp:55-92): one perform_registration per iteration -- answered
by the resident server behind ws_reg_iterate (round 6) -- and the oracle's host update (wso_gn_update) in between; every
iteration's h, g, e, c and the final pose must be the oracle's bit for bit.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pose_error(A, B):
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    dt = np.linalg.norm(A[:3, 3] - B[:3, 3]) / 1000.0
    R = A[:3, :3] @ B[:3, :3].T
    k = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    return dt, float(np.arctan2(np.linalg.norm(k), (np.trace(R) - 1) / 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--server", action="store_true")
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    rng = np.random.default_rng(args.seed)
    bad = exact = 0
    for case in range(args.cases):
        res = int(rng.choice([32, 50, 64]))
        tau = int(rng.choice([600, 1000]))
        size = (int(rng.integers(50, 80)) * 2, int(rng.integers(50, 80)) * 2, int(rng.integers(24, 40)) * 2)
        mw = 640
        params = W.Params(W.MapParams(resolution=res, max_distance=tau / 1000.0, max_weight=mw // 64, size=tuple(s * res / 1000.0 for s in size)))
        lm = W.LocalMap(size[0], size[1], size[2], tau, 0)
        reg = W.TSDFRegistration(params, lm)
        oa = O.OracleMap(size, tau, 0)
        on = oa.copy()
        he = tuple(np.array(size) * res * rng.uniform(0.3, 0.42, 3))
        pts = None
        for k in range(int(rng.integers(1, 3))):
            pts = S.os1_128_scan(rings=int(rng.choice([32, 64])), azimuths=int(rng.choice([256, 512])), half_extents_mm=he, seed=int(rng.integers(1, 1 << 30)))
            O.update_tsdf(oa, on, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
            reg.update_tsdf(torch.from_numpy(pts).cuda(), pose=np.eye(4, dtype=np.float32))
        Tp = S.perturbation(float(rng.uniform(-120, 120)), float(rng.uniform(-120, 120)), float(rng.uniform(-20, 20)), float(rng.uniform(-6, 6)))
        q = S.transform_points_mm(pts, Tp)
        max_it = int(rng.choice([200, 200, 60]))
        rp = reg.params_.registration
        rp.max_iterations = max_it
        T_cpu, it_cpu, _ = O.register_cloud(oa, q, np.eye(4), max_it, rp.it_weight_gradient, rp.epsilon, res)
        if args.server:
            import ctypes as C
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from test_abi_and_host import OracleGnBackend
            st = OracleGnBackend.State()
            O.lib().wso_gn_begin(C.byref(st), O._p(O.colmajor(np.eye(4, dtype=np.float32))), int(max_it), C.c_float(rp.it_weight_gradient), C.c_float(rp.epsilon))
            reg.reg_.prepare_registration(q)
            its = 0
            while not (st.finished or st.iterations >= st.max_iterations):
                T = np.ctypeslib.as_array(st.T).reshape(4, 4).T.copy()
                h, g, e, c = reg.reg_.perform_registration(reg.tsdf().device_map(), T, res)
                ho, go, eo, co = O.reg_iterate(oa, T, q, res, reg.reg_.flags)
                if not (c == co and e == eo and np.array_equal(g, go) and np.array_equal(h, ho)):
                    print(f"case {case}: sums differ at iteration {its}")
                    bad += 1
                    break
                sums = np.concatenate([h.T.reshape(-1), g, [e, c]]).astype(np.int64)
                O.lib().wso_gn_update(C.byref(st), O._p(np.ascontiguousarray(sums)))
                its += 1
            T_gpu = np.ctypeslib.as_array(st.T).reshape(4, 4).T.copy()
            reg.last_iterations = int(st.iterations)
        else:
            T_gpu = reg.register_cloud(q, np.eye(4, dtype=np.float32))
        dt, ang = pose_error(T_gpu, T_cpu)
        same = np.array_equal(np.asarray(T_gpu, dtype=np.float32), np.asarray(T_cpu, dtype=np.float32))
        exact += int(same)
        ok = reg.last_iterations == it_cpu and dt < 1e-4 and ang < 1e-4
        bad += 0 if ok else 1
        print(f"case {case:3d}: res {res} tau {tau} size {size} points {len(q)} iterations gpu {reg.last_iterations} oracle {it_cpu} "
              f"dt {dt:.2e} m dang {ang:.2e} rad bit-identical {same} -> {'ok' if ok else 'MISMATCH'}")
    print(f"{args.cases} cases, {bad} mismatches, {exact} bit-identical poses")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
