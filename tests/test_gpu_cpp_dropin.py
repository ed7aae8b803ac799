"""The C++ drop-in classes (include/warpsense_hip/{compat,mapping}.hpp) driven by examples/harness.cpp:
same call sequence as the reference's test/pcd_registration.cpp, result compared with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from warpsense_amd import synthetic as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_harness_matches_oracle(tmp_path):
    exe = os.path.join(ROOT, "examples", "harness")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    tau, res, mw, edge = 1000, 50, 640, 96
    pts = S.os1_128_scan(rings=32, azimuths=256, half_extents_mm=(2000.0, 1700.0, 800.0), seed=2)
    pert = S.transform_points_mm(pts, S.perturbation(35, -20, 8, 1.5))
    pts.tofile(tmp_path / "scan.bin")
    pert.tofile(tmp_path / "pert.bin")
    out = subprocess.run([exe, str(tmp_path / "scan.bin"), str(tmp_path / "pert.bin"), str(len(pts)), str(edge), str(res),
                          str(tau), str(mw), str(tmp_path / "avg.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    iters = int(lines[0].split()[1])
    T = np.array([float(x) for x in lines[1].split()], dtype=np.float32).reshape(4, 4).T

    oa = O.OracleMap((edge, edge, edge), tau, 0)
    on = oa.copy()
    O.update_tsdf(oa, on, pts, (0, 0, 0), (0, 0, 32768), tau, mw, res)
    avg = np.fromfile(tmp_path / "avg.bin", dtype=np.uint32)
    assert np.array_equal(avg, oa.data)
    To, ito, _ = O.register_cloud(oa, pert, np.eye(4), 200, 0.1, 0.03, res)
    assert iters == ito
    assert np.abs(T - To).max() < 1e-4
