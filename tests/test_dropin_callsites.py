"""The drop-in boundary compiles the reference's own call sites: tests/cpp/callsites.cpp holds the
statements of tsdf_mapping.cpp:114,141, app.cpp:218, featsense/mapping.cpp:187, pcd2tsdf.cpp:117 verbatim in meaning and
is compiled through the forwarding headers this repo ships under the reference's own names
(include/warpsense/cuda/{update_tsdf,registration,device_map,device_map_wrapper,cleanup}.h -> warpsense_hip/compat.hpp), once
with compat.hpp's own POD math types (only -I<repo>/include) and once exactly as INTEGRATION.md says a maintainer builds:
-I<repo>/include AHEAD of -I<reference>/include, so the names resolve to the forwarding headers and the rmagine:: math types
and TSDFEntry are the reference's own (only where /root/reference exists)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "callsites.cpp")
REF_INC = "/root/reference/include"


def _compile(extra):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    cmd = [cxx, "-std=c++17", "-fsyntax-only", "-Wall", f"-I{os.path.join(ROOT, 'include')}", *extra, SRC]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode(errors="replace")


def test_callsites_compile_with_own_types():
    _compile([])


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="reference tree not present on this box")
def test_callsites_compile_with_reference_math_types():
    _compile([f"-I{REF_INC}", "-DCALLSITES_EXPECT_REFERENCE_TYPES"])


def test_forwarding_headers_exist_for_every_device_header_the_callers_include():
    for name in ("update_tsdf", "registration", "device_map", "device_map_wrapper", "cleanup"):
        path = os.path.join(ROOT, "include", "warpsense", "cuda", name + ".h")
        text = open(path).read()
        assert '#include "warpsense_hip/compat.hpp"' in text and "#pragma once" in text


def test_buffer_bounds_follow_the_reference_size_t_semantics(tmp_path):
    """in_bounds_with_buffer_{neg,pos} (device_map.h:116-128) compare as size_t: checked against the oracle's restatement
    (itself pinned against the reference header through oracle/_ref) on random points incl. buffer > size/2."""
    import oracle_lib as O
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    prog = tmp_path / "bounds.cpp"
    prog.write_text(r'''
#include <cstdio>
#include "warpsense_hip/compat.hpp"
int main(int argc, char **argv)
{
  int size[3] = {21, 15, 9}, off[3] = {10, 7, 4}, pos[3] = {3, -2, 1};
  cuda::DeviceMap m(size, off, nullptr, pos);
  int x, y, z; unsigned long b;
  while (scanf("%d %d %d %lu", &x, &y, &z, &b) == 4)
    printf("%d %d\n", (int)m.in_bounds_with_buffer_neg(rmagine::Vector3i(x, y, z), b), (int)m.in_bounds_with_buffer_pos(rmagine::Vector3i(x, y, z), b));
  return 0;
}
''')
    exe = tmp_path / "bounds"
    subprocess.check_call([cxx, "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", str(prog), "-o", str(exe)])
    rng = np.random.default_rng(0)
    q = np.concatenate([rng.integers(-20, 21, (400, 3)), rng.integers(0, 13, (400, 1))], axis=1)
    out = subprocess.run([str(exe)], input="\n".join(" ".join(str(int(v)) for v in r) for r in q).encode(), stdout=subprocess.PIPE, check=True)
    got = np.array([[int(v) for v in l.split()] for l in out.stdout.decode().splitlines()])
    om = O.OracleMap((21, 15, 9), 1000, 0, pos=(3, -2, 1), offset=(10, 7, 4))
    v = om.view()
    import ctypes as C
    for (x, y, z, b), (gn, gp) in zip(q, got):
        assert gn == O.lib().wso_in_bounds_with_buffer_neg(C.byref(v), int(x), int(y), int(z), int(b)), (x, y, z, b)
        assert gp == O.lib().wso_in_bounds_with_buffer_pos(C.byref(v), int(x), int(y), int(z), int(b)), (x, y, z, b)
