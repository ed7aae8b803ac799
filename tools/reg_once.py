"""Two registrations of the benchmark cloud against the 513^3 map -- with a -DWS_REG_TIMING build of the library the
resident loop prints its per-phase ticks (exchange / solve / accumulate / reduce / arrive) for a sample of workgroups:

    python -m warpsense_amd.build --variant regtiming "-DWS_REG_TIMING"
    WS_HIP_LIB=$PWD/warpsense_amd/variants/regtiming.so python tools/reg_once.py
"""
import os, sys
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
for _ in range(2):
    T, it = reg.register_cloud(tsdf.device_map(), eye, 200, 0.1, 0.03, res)
torch.cuda.synchronize()
print("iterations", it)
