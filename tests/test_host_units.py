"""Host-side units of the record key / voxel-byte layout (ws_internal.h) and of the walk over column changes (ws_dda.h), built and
run on the CPU: no GPU needed (hipcc cross-compiles; only host code runs)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "warpsense_amd", "csrc")


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_record_key_split_and_brick_order(tmp_path):
    """rec_format / make_rec / rec_negative and vbrick: the per-scan split of the 38 key bits keeps records ascending in
    (point, ray step, fan step) and recoverable; the brick order is a bijection of a tile's 1024 voxels that keeps a thread's four
    consecutive z in one aligned word."""
    if not os.path.exists(_hipcc()):
        pytest.skip("hipcc not installed")
    exe = tmp_path / "host_units"
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O1", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
                           os.path.join(ROOT, "tests", "cpp", "host_units.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def test_column_change_walk_equals_the_sample_walk(tmp_path):
    """ws_dda.h (what the free pass walks rays with since round 5) against the reference's sample-by-sample loop
    (update_tsdf.cu:65-76) on random rays: same emitting steps, same sample positions -- through the cell around zero, with
    zero components, on any number of lanes per ray.  (tools/dda_check.cpp; 300 k rays when run by hand.)"""
    exe = tmp_path / "dda_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", f"-I{CSRC}", os.path.join(ROOT, "tools", "dda_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "25000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok:"), out.stdout + out.stderr
