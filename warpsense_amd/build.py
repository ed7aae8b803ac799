"""In-tree build of libwarpsense_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m warpsense_amd.build            # build if sources are newer than the library
    python -m warpsense_amd.build --force

The library is written to warpsense_amd/libwarpsense_hip.so (git-ignored, travels with gpurun snapshots).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libwarpsense_hip.so")
SOURCES = ["api.hip", "tsdf_update.hip", "tsdf_tiles.hip", "registration.hip"]
HEADERS = [os.path.join(CSRC, "ws_internal.h"), os.path.join(CSRC, "ws_device.h"), os.path.join(CSRC, "ws_march.h"),
           os.path.join(CSRC, "ws_tiles.h"),
           os.path.join(ROOT, "include", "warpsense_hip.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def flags() -> list[str]:
    # -ffp-contract=off: the Gauss-Newton update must round like the host code of the reference (no FMA fusion);
    # correctly rounded sqrt/div are hipcc's default and are required by the integer truncations.
    return [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-Wall", "-Wno-unused-function", *os.environ.get("WS_EXTRA_FLAGS", "").split(), f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    objs = []
    procs = []
    os.makedirs(os.path.join(PKG_DIR, "build"), exist_ok=True)
    for s in SOURCES:
        obj = os.path.join(PKG_DIR, "build", s.replace(".hip", ".o"))
        cmd = [hipcc(), *flags(), "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out.decode(errors="replace"))
        if verbose and out:
            print(out.decode(errors="replace"))
    link = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
