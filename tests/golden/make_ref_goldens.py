#!/usr/bin/env python3
"""Generate tests/golden/ref_headers.npz from oracle/_ref (the REFERENCE'S OWN headers compiled by g++).

Run in the build container (needs /root/reference):   make -C oracle ref && python tests/golden/make_ref_goldens.py

The fixture holds inputs and the reference's outputs only (data, no reference source):
  ring-buffer index math        cuda::DeviceMap::{get_index,in_bounds,in_bounds_with_buffer_pos/neg}
  vector math                   rmagine::Vector3<int|long>::l2norm / cross and the ray set-up expressions
                                of update_tsdf.cu:57-63 evaluated through vector3.h's operators
  layouts                       TSDFEntry packing, Matrix4x4f / Matrix6x6l storage order, consts
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    R = O.ref_lib()
    if R is None:
        raise SystemExit("oracle/_ref/libws_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20260929)
    out = {}

    # ---- ring buffer
    maps, queries, results = [], [], []
    for _ in range(40):
        size = rng.integers(1, 40, 3) * 2 + 1
        pos = rng.integers(-500, 500, 3)
        offset = np.array([rng.integers(0, s) for s in size])
        data = np.zeros(int(np.prod(size)), dtype=np.uint32)
        s32, p32, o32 = size.astype(np.int32), pos.astype(np.int32), offset.astype(np.int32)
        h = R.ref_map_create(s32.ctypes.data, p32.ctypes.data, o32.ctypes.data, data.ctypes.data)
        for _ in range(64):
            q = pos + rng.integers(-(size // 2) - 3, size // 2 + 4)
            buf = int(rng.integers(0, 3))
            inb = R.ref_in_bounds(h, int(q[0]), int(q[1]), int(q[2]))
            idx = R.ref_get_index(h, int(q[0]), int(q[1]), int(q[2])) if inb else -1
            results.append([inb, idx, R.ref_in_bounds_pos(h, int(q[0]), int(q[1]), int(q[2]), buf),
                            R.ref_in_bounds_neg(h, int(q[0]), int(q[1]), int(q[2]), buf), buf])
            queries.append(q)
            maps.append(np.concatenate([size, pos, offset]))
        R.ref_map_destroy(h)
    out["ring_maps"] = np.array(maps, dtype=np.int32)
    out["ring_queries"] = np.array(queries, dtype=np.int32)
    out["ring_results"] = np.array(results, dtype=np.int64)

    # ---- vector math
    v = rng.integers(-20000, 20000, (2000, 3)).astype(np.int32)
    v[:50] = rng.integers(-40000, 40000, (50, 3))  # int32 wrap in the squared norm
    out["l2_in"] = v
    out["l2_i"] = np.array([R.ref_l2norm_i(int(a), int(b), int(c)) for a, b, c in v], dtype=np.int32)
    vl = rng.integers(-2**31, 2**31, (2000, 3)).astype(np.int64)
    out["l2l_in"] = vl
    out["l2_l"] = np.array([R.ref_l2norm_l(int(a), int(b), int(c)) for a, b, c in vl], dtype=np.int64)
    a = rng.integers(-30000, 30000, (500, 3)).astype(np.int32)
    b = rng.integers(-30000, 30000, (500, 3)).astype(np.int32)
    cr = np.zeros((500, 3), dtype=np.int32)
    for i in range(500):
        R.ref_cross_i(a[i].ctypes.data, b[i].ctypes.data, cr[i].ctypes.data)
    out["cross_a"], out["cross_b"], out["cross_out"] = a, b, cr

    # ---- ray set-up (update_tsdf.cu:57-63 through vector3.h)
    pts = rng.integers(-15000, 15000, (1500, 3)).astype(np.int32)
    pos_mm = rng.integers(-300, 300, (1500, 3)).astype(np.int32) * 50 + 25
    ups = np.tile(np.array([0, 0, 32768], dtype=np.int32), (1500, 1))
    ups[500:] = rng.integers(-32768, 32768, (1000, 3))
    dist = np.zeros(1500, dtype=np.int32)
    interp = np.zeros((1500, 3), dtype=np.int64)
    rc = np.zeros(1500, dtype=np.int32)
    for i in range(1500):
        d = C.c_int32(0)
        rc[i] = R.ref_ray_setup(pts[i].ctypes.data, pos_mm[i].ctypes.data, ups[i].ctypes.data, C.byref(d), interp[i].ctypes.data)
        dist[i] = d.value
    out["ray_points"], out["ray_pos"], out["ray_up"] = pts, pos_mm, ups
    out["ray_distance"], out["ray_interp"], out["ray_rc"] = dist, interp, rc

    # ---- layouts
    vw = rng.integers(-32768, 32768, (200, 2)).astype(np.int16)
    out["entry_vw"] = vw
    out["entry_raw"] = np.array([R.ref_pack(int(x), int(y)) for x, y in vw], dtype=np.uint32)
    m4 = np.zeros(16, dtype=np.float32)
    R.ref_matrix4_layout(m4.ctypes.data)
    m6 = np.zeros(36, dtype=np.int64)
    R.ref_matrix6_layout(m6.ctypes.data)
    out["m4_layout"], out["m6_layout"] = m4, m6
    out["consts"] = np.array([R.ref_consts(0), R.ref_consts(1), R.ref_sizeof_entry(), R.ref_sizeof_long()], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "ref_headers.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_headers.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
