"""How much of a Gauss-Newton iteration of the resident loop is the gathers of points that changed voxel?  The benchmark
registration (178 iterations from the perturbed pose) against the same loop started AT the converged pose with epsilon < 0
(never stops: 200 iterations in which next to no point moves).

    python tools/reg_converged.py
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import warpsense_amd as W
from warpsense_amd import synthetic as S
tau, res = 1000, 50
view = W.DeviceMap([513, 513, 513], [256, 256, 256], None, (0, 0, 0))
tsdf = W.TSDFCuda(view, tau, 640, res)
pts = S.os1_128_scan()
tsdf.update_tsdf(torch.from_numpy(pts).cuda(), (0, 0, 0), (0, 0, 32768))
reg = W.RegistrationCuda(None)
q = torch.from_numpy(S.transform_points_mm(pts, S.perturbation(100, 100, 0, 5.0))).cuda()
reg.prepare_registration(q)
eye = np.eye(4, dtype=np.float32)


def timed(T_in, max_it, eps, reps=20):
    for _ in range(3):
        T, it = reg.register_cloud(tsdf.device_map(), T_in, max_it, 0.1, eps, res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        T, it = reg.register_cloud(tsdf.device_map(), T_in, max_it, 0.1, eps, res)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return T, it, dt


T, it, dt = timed(eye, 200, 0.03)
print(f"from the perturbed pose: {it} iterations, {dt * 1e6:.1f} us, {dt * 1e6 / it:.3f} us per iteration")
T2, it2, dt2 = timed(T, 200, -1.0)
print(f"from the converged pose, never stopping: {it2} iterations, {dt2 * 1e6:.1f} us, {dt2 * 1e6 / it2:.3f} us per iteration")
T3, it3, dt3 = timed(eye, 200, -1.0)
print(f"from the perturbed pose, never stopping: {it3} iterations, {dt3 * 1e6:.1f} us, {dt3 * 1e6 / it3:.3f} us per iteration")
