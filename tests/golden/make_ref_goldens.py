#!/usr/bin/env python3
"""Generate tests/golden/ref_headers.npz from oracle/_ref (the REFERENCE'S OWN headers compiled by g++).

Run in the build container (needs /root/reference):   make -C oracle ref && python tests/golden/make_ref_goldens.py

The fixture holds inputs and the reference's outputs only (data, no reference source):
  ring-buffer index math        cuda::DeviceMap::{get_index,in_bounds,in_bounds_with_buffer_pos/neg}
  vector math                   rmagine::Vector3<int|long>::l2norm / cross and the ray set-up expressions
                                of update_tsdf.cu:57-63 evaluated through vector3.h's operators
  layouts                       TSDFEntry packing, Matrix4x4f / Matrix6x6l storage order, consts
  integrate                     cu_avg_tsdf_krnl's per-voxel body (update_tsdf.cu:19-41) through TSDFEntry's accessors, 12 000 pairs
  Jacobians                     calc_jacobis_krnl's lookups / gradient rule / cross product (registration.cu:217-253) through
                                cuda::DeviceMap::value_unchecked and Vector3::cross, 24 maps x 500 points
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

    # ---- cu_avg_tsdf_krnl's per-voxel body through TSDFEntry's accessors (update_tsdf.cu:19-41)
    n_int = 12000
    tau_i, mw_i = 600, 640
    ex = np.stack([rng.integers(-tau_i, tau_i + 1, n_int), rng.integers(-80, 700, n_int)], axis=1).astype(np.int16)
    fr = np.stack([rng.integers(-tau_i, tau_i + 1, n_int), rng.integers(-80, 80, n_int)], axis=1).astype(np.int16)
    # int16 wrap of the stored average / weight sum, extreme values and weights, untouched and negative-weight cases
    ex[:2000] = rng.integers(-32768, 32768, (2000, 2))
    fr[:2000] = rng.integers(-32768, 32768, (2000, 2))
    fr[2000:2600, 1] = 0
    ex[2600:3200, 1] = 0
    ex[3200:3800, 1] = rng.integers(-64, 1, 600)
    fr[3800:4400, 1] = rng.integers(-64, 0, 600)
    ex[4400:5000] = np.stack([np.full(600, tau_i), np.zeros(600)], axis=1)  # the default entry
    ex_raw = np.array([R.ref_pack(int(v), int(w)) for v, w in ex], dtype=np.uint32)
    fr_raw = np.array([R.ref_pack(int(v), int(w)) for v, w in fr], dtype=np.uint32)
    ex_out = np.zeros(n_int, dtype=np.uint32)
    fr_out = np.zeros(n_int, dtype=np.uint32)
    for i in range(n_int):
        f = C.c_uint32(int(fr_raw[i]))
        ex_out[i] = R.ref_integrate_entry(int(ex_raw[i]), C.byref(f), mw_i, tau_i)
        fr_out[i] = f.value
    out["avg_existing"], out["avg_fresh"] = ex_raw, fr_raw
    out["avg_existing_out"], out["avg_fresh_out"] = ex_out, fr_out
    out["avg_params"] = np.array([mw_i, tau_i], dtype=np.int32)

    # ---- calc_jacobis_krnl's lookups, gradient rule and cross product (registration.cu:217-253) on maps of random entries.
    # The kernel's fixed-point transform is exact for a pure integer translation t (M = 32768 I, q = p + t, centre = t), so
    # (buf, point) = (trunc((p + t) / res), p) are what it hands to this half; `jac_T` lets the oracle be driven the same way.
    jm, jpts, jT, jres_, jout = [], [], [], [], []
    for case in range(24):
        size = rng.integers(4, 12, 3) * 2 + 1
        pos = rng.integers(-40, 40, 3)
        offset = np.array([rng.integers(0, s) for s in size])
        n_v = int(np.prod(size))
        vals = rng.integers(-600, 601, n_v)
        wts = rng.integers(-20, 60, n_v)
        wts[rng.random(n_v) < 0.25] = 0           # unobserved voxels
        vals[rng.random(n_v) < 0.10] = 0          # a zero on one side is not "opposite sign"
        if case % 4 == 3:                         # arbitrary raw entries: int32 wrap in the cross product
            vals = rng.integers(-32768, 32768, n_v)
        data = ((vals.astype(np.int64) & 0xffff) | ((wts.astype(np.int64) & 0xffff) << 16)).astype(np.uint32)
        res = int(rng.choice([20, 50, 64]))
        t = rng.integers(-3, 4, 3) * res + rng.integers(0, res, 3)
        s32, p32, o32 = size.astype(np.int32), pos.astype(np.int32), offset.astype(np.int32)
        h = R.ref_map_create(s32.ctypes.data, p32.ctypes.data, o32.ctypes.data, data.ctypes.data)
        lo = (pos - size // 2 - 2) * res
        hi = (pos + size // 2 + 3) * res
        q = rng.integers(lo, hi, (500, 3))        # transformed points, a margin outside the window included
        if case % 4 == 3:
            q[:100] += rng.integers(-40000, 40000, (100, 3)) * 0  # (kept inside: wrap comes from the entries)
        pts_c = (q - t).astype(np.int32)           # what the kernel is given
        buf = np.trunc(q / res).astype(np.int32)   # C truncation of q / res
        assert np.array_equal(buf, np.where(q >= 0, q // res, -((-q) // res)))
        rows = np.zeros((500, 8), dtype=np.int64)
        for i in range(500):
            J = np.zeros(6, dtype=np.int64)
            v = C.c_int16(0)
            pnt = pts_c[i].copy()
            mk = R.ref_jacobi(h, buf[i].ctypes.data, pnt.ctypes.data, J.ctypes.data, C.byref(v))
            rows[i, 0] = mk
            if mk:
                rows[i, 1] = v.value
                rows[i, 2:] = J
        R.ref_map_destroy(h)
        jm.append(np.concatenate([size, pos, offset, [res], t]))
        jpts.append(pts_c)
        jout.append(rows)
        out[f"jac_data_{case}"] = data
    out["jac_maps"] = np.array(jm, dtype=np.int32)      # size(3) pos(3) offset(3) res t(3)
    out["jac_points"] = np.array(jpts, dtype=np.int32)  # [case][500][3]
    out["jac_out"] = np.array(jout, dtype=np.int64)     # [case][500][mask, value, J(6)]

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
