"""In-tree build of libwarpsense_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m warpsense_amd.build            # build if sources are newer than the library
    python -m warpsense_amd.build --force

The library is written to warpsense_amd/libwarpsense_hip.so (git-ignored, travels with gpurun snapshots).

WS_EXTRA_FLAGS adds compiler flags, e.g. the instrumentation switches of the kernels (they print per-phase clock
ticks from a few workgroups; never used in the shipped build):
    -DWS_REG_TIMING / -DWS_REG_TIMING_GN   phases of the resident registration loop / of the Gauss-Newton update
    -DWS_REG_BLOCKS=.. -DWS_REG_THREADS=..  grid shape of the registration kernels (default 256 x 512)
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
H5_LIB_PATH = os.path.join(PKG_DIR, "libwarpsense_h5.so")  # optional: global-map file (needs the HDF5 C library)
SOURCES = ["api.hip", "tsdf_update.hip", "tsdf_integrate.hip", "registration.hip", "scan_preprocess.hip"]
HEADERS = [os.path.join(CSRC, "ws_internal.h"), os.path.join(CSRC, "ws_device.h"), os.path.join(CSRC, "ws_march.h"), os.path.join(CSRC, "ws_dda.h"),
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


def build_variant(name: str, extra_flags: str, verbose: bool = False) -> str:
    """warpsense_amd/variants/<name>.so: the library built with extra compiler flags (kernel tuning switches), selected at
    run time with WS_HIP_LIB=<path> -- several variants measured back to back on ONE box (the boxes differ by a few %)."""
    out = os.path.join(PKG_DIR, "variants", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    old = os.environ.get("WS_EXTRA_FLAGS", "")
    os.environ["WS_EXTRA_FLAGS"] = (old + " " + extra_flags).strip()
    try:
        return build_native(force=True, verbose=verbose, lib_path=out, obj_dir=os.path.join(PKG_DIR, "build", "variant_" + name))
    finally:
        os.environ["WS_EXTRA_FLAGS"] = old


def build_native(force: bool = False, verbose: bool = False, lib_path: str = LIB_PATH, obj_dir: str | None = None) -> str:
    if not force and not needs_build():
        return lib_path
    objs = []
    procs = []
    obj_dir = obj_dir or os.path.join(PKG_DIR, "build")
    os.makedirs(obj_dir, exist_ok=True)
    for s in SOURCES:
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
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
    link = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", lib_path]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return lib_path


def find_hdf5() -> tuple[str, str] | None:
    """(include dir, lib dir) of an installed HDF5 C library, or None."""
    roots = [os.environ.get("HDF5_ROOT"), "/opt/conda", "/usr", "/usr/local"]
    layouts = [("include", "lib"), ("include/hdf5/serial", "lib/x86_64-linux-gnu/hdf5/serial"), ("include", "lib/x86_64-linux-gnu"),
               ("include", "lib64")]
    for root in roots:
        if not root:
            continue
        for inc, lib in layouts:
            i, l = os.path.join(root, inc), os.path.join(root, lib)
            if os.path.exists(os.path.join(i, "hdf5.h")) and os.path.exists(os.path.join(l, "libhdf5.so")):
                return i, l
    return None


def build_h5(force: bool = False, verbose: bool = False) -> str | None:
    """libwarpsense_h5.so (include/warpsense_h5.h) with the host C++ compiler; None when HDF5 is not installed."""
    src = os.path.join(CSRC, "ws_h5.cpp")
    hdr = os.path.join(ROOT, "include", "warpsense_h5.h")
    found = find_hdf5()
    if found is None:
        return None
    if not force and os.path.exists(H5_LIB_PATH) and os.path.getmtime(H5_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return H5_LIB_PATH
    inc, lib = found
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        return None
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", f"-I{os.path.join(ROOT, 'include')}", f"-I{inc}", src, "-o", H5_LIB_PATH,
           f"-L{lib}", "-lhdf5", f"-Wl,-rpath,{lib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("building libwarpsense_h5.so failed:\n" + r.stdout.decode(errors="replace"))
    return H5_LIB_PATH


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m warpsense_amd.build --variant NAME "-DFLAG=1 ..."
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2] if len(sys.argv) > i + 2 else "", verbose=True))
        sys.exit(0)
    print(build_native(force="--force" in sys.argv, verbose=True))
    print(build_h5(force="--force" in sys.argv, verbose=True) or "libwarpsense_h5.so: skipped (no HDF5 C library found)")
