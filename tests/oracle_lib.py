"""ctypes bindings for the CPU oracle (oracle/libws_oracle.so), the CPU-baseline port and oracle/_ref.

TEST INFRASTRUCTURE ONLY — nothing under warpsense_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

REG_ALL_POINTS = 0
REG_COMPAT_REFERENCE_LAUNCH = 1


def _ensure_built():
    so = os.path.join(ORACLE_DIR, "libws_oracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("ws_oracle.c", "ws_oracle.h", "cpu_baseline.cpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return so


class WsoMap(C.Structure):
    _fields_ = [("size", C.c_int32 * 3), ("pos", C.c_int32 * 3), ("offset", C.c_int32 * 3),
                ("data", C.POINTER(C.c_uint32))]


class WsoStats(C.Structure):
    _fields_ = [("write_calls", C.c_int64), ("accepted", C.c_int64), ("rays_in_bounds", C.c_int64),
                ("rays_degenerate", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_ensure_built())
        L = _lib
        P = C.POINTER
        L.wso_get_index.restype = C.c_int64
        L.wso_get_index.argtypes = [P(WsoMap), C.c_int32, C.c_int32, C.c_int32]
        for name in ("wso_in_bounds",):
            getattr(L, name).argtypes = [P(WsoMap), C.c_int32, C.c_int32, C.c_int32]
        for name in ("wso_in_bounds_with_buffer_pos", "wso_in_bounds_with_buffer_neg"):
            getattr(L, name).argtypes = [P(WsoMap), C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.wso_pack.restype = C.c_uint32
        L.wso_pack.argtypes = [C.c_int16, C.c_int16]
        L.wso_tsdf_min.argtypes = [P(C.c_uint32), C.c_uint32]
        L.wso_dz_per_distance.restype = C.c_int32
        L.wso_update_min.argtypes = [P(WsoMap), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, P(WsoStats)]
        L.wso_update_avg.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32]
        L.wso_update_tsdf.argtypes = [P(WsoMap), P(WsoMap), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_int32, C.c_int32, P(WsoStats)]
        L.wso_calc_jacobis.argtypes = [P(WsoMap), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_uint32]
        L.wso_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                 P(C.c_int32), P(C.c_int32), C.c_uint32]
        L.wso_reg_iterate.argtypes = [P(WsoMap), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p,
                                      C.c_void_p, P(C.c_int32), P(C.c_int32), C.c_uint32]
        L.wso_register_cloud.argtypes = [P(WsoMap), C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_float,
                                         C.c_float, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32]
        L.wso_register_cloud.restype = C.c_int
        L.wso_convert_pose.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.wso_to_int_mat.argtypes = [C.c_void_p, C.c_void_p]
        L.wso_transform_point.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wso_to_map.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.wso_xi_to_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wso_solve6.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wso_preprocess.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]
        L.wso_preprocess.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(value, weight):
    v = np.asarray(value).astype(np.int16).astype(np.uint16).astype(np.uint32)
    w = np.asarray(weight).astype(np.int16).astype(np.uint16).astype(np.uint32)
    return v | (w << np.uint32(16))


def unpack(raw):
    raw = np.asarray(raw, dtype=np.uint32)
    return (raw & 0xFFFF).astype(np.uint16).astype(np.int16), (raw >> 16).astype(np.uint16).astype(np.int16)


class OracleMap:
    """Host ring-buffer local map with the reference's construction rules
    (sizes forced odd, offset = size/2, filled with the default entry; src/map/hdf5_local_map.cpp:5-20)."""

    def __init__(self, size, default_value, default_weight=0, pos=(0, 0, 0), offset=None, data=None):
        size = [int(s) if int(s) % 2 == 1 else int(s) + 1 for s in size]
        self.size = np.array(size, dtype=np.int32)
        self.pos = np.array(pos, dtype=np.int32)
        self.offset = np.array(offset if offset is not None else [s // 2 for s in size], dtype=np.int32)
        n = int(np.prod(self.size.astype(np.int64)))
        if data is None:
            self.data = np.full(n, pack(default_value, default_weight), dtype=np.uint32)
        else:
            self.data = np.ascontiguousarray(data, dtype=np.uint32)
            assert self.data.size == n

    @property
    def n_vox(self):
        return int(self.data.size)

    def copy(self):
        return OracleMap(self.size, 0, 0, self.pos, self.offset, self.data.copy())

    def view(self):
        m = WsoMap()
        for k in range(3):
            m.size[k], m.pos[k], m.offset[k] = int(self.size[k]), int(self.pos[k]), int(self.offset[k])
        m.data = self.data.ctypes.data_as(C.POINTER(C.c_uint32))
        return m

    def index(self, x, y, z):
        v = self.view()
        return int(lib().wso_get_index(C.byref(v), x, y, z))

    def in_bounds(self, x, y, z):
        v = self.view()
        return bool(lib().wso_in_bounds(C.byref(v), x, y, z))

    def entry(self, x, y, z):
        v, w = unpack(self.data[self.index(x, y, z)])
        return int(v), int(w)

    def set_entry(self, x, y, z, value, weight):
        self.data[self.index(x, y, z)] = pack(value, weight)


def convert_pose(pose_rowmajor, res):
    """pose: 4x4 numpy (math layout). Returns (pos_vox[3], up[3]) like TSDFMapping::convert_pose_to_gpu."""
    T = np.ascontiguousarray(np.asarray(pose_rowmajor, dtype=np.float32).T)  # -> column-major flat
    pos = np.zeros(3, dtype=np.int32)
    up = np.zeros(3, dtype=np.int32)
    lib().wso_convert_pose(_p(T), int(res), _p(pos), _p(up))
    return pos, up


def update_tsdf(avg: OracleMap, new: OracleMap, points, scanner_pos, up, tau, max_weight, res):
    pts = np.ascontiguousarray(points, dtype=np.int32)
    sp = np.ascontiguousarray(scanner_pos, dtype=np.int32)
    u = np.ascontiguousarray(up, dtype=np.int32)
    st = WsoStats()
    av, nv = avg.view(), new.view()
    lib().wso_update_tsdf(C.byref(av), C.byref(nv), _p(pts), pts.shape[0], _p(sp), _p(u), int(tau), int(max_weight),
                          int(res), C.byref(st))
    return st


def update_min(new: OracleMap, points, scanner_pos, up, tau, res):
    pts = np.ascontiguousarray(points, dtype=np.int32)
    sp = np.ascontiguousarray(scanner_pos, dtype=np.int32)
    u = np.ascontiguousarray(up, dtype=np.int32)
    st = WsoStats()
    nv = new.view()
    lib().wso_update_min(C.byref(nv), _p(pts), pts.shape[0], _p(sp), _p(u), int(tau), int(res), C.byref(st))
    return st


def update_avg(new: OracleMap, avg: OracleMap, max_weight, tau):
    lib().wso_update_avg(_p(new.data), _p(avg.data), new.n_vox, int(max_weight), int(tau))


def colmajor(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(-1)


def reg_iterate(m: OracleMap, T_rowmajor, points, res, flags=REG_ALL_POINTS):
    pts = np.ascontiguousarray(points, dtype=np.int32)
    T = colmajor(T_rowmajor)
    h = np.zeros(36, dtype=np.int64)
    g = np.zeros(6, dtype=np.int64)
    e, c = C.c_int32(0), C.c_int32(0)
    v = m.view()
    lib().wso_reg_iterate(C.byref(v), _p(T), _p(pts), pts.shape[0], int(res), _p(h), _p(g), C.byref(e), C.byref(c),
                          int(flags))
    return h.reshape(6, 6).T.copy(), g, e.value, c.value  # h returned in math layout h[i, j]


def calc_jacobis(m: OracleMap, T_rowmajor, points, res, flags=REG_ALL_POINTS):
    pts = np.ascontiguousarray(points, dtype=np.int32)
    n = pts.shape[0]
    T = colmajor(T_rowmajor)
    J = np.zeros((n, 6), dtype=np.int64)
    vals = np.zeros(n, dtype=np.int16)
    mask = np.zeros(n, dtype=np.uint8)
    v = m.view()
    lib().wso_calc_jacobis(C.byref(v), _p(T), _p(pts), n, int(res), _p(J), _p(vals), _p(mask), int(flags))
    return J, vals, mask


def register_cloud(m: OracleMap, points, T_in_rowmajor, max_iterations=200, it_weight_gradient=0.1, epsilon=0.03,
                   res=50, flags=REG_ALL_POINTS, trace_cap=0):
    pts = np.ascontiguousarray(points, dtype=np.int32)
    T = colmajor(T_in_rowmajor)
    out = np.zeros(16, dtype=np.float32)
    trace = np.zeros((max(trace_cap, 1), 44), dtype=np.int64)
    v = m.view()
    it = lib().wso_register_cloud(C.byref(v), _p(pts), pts.shape[0], _p(T), int(max_iterations),
                                  C.c_float(it_weight_gradient), C.c_float(epsilon), int(res), int(flags), _p(out),
                                  _p(trace) if trace_cap else None, int(trace_cap))
    return out.reshape(4, 4).T.copy(), it, trace[:min(it, trace_cap)]


# ---------------------------------------------------------------- CPU baseline port (oracle/cpu_baseline.cpp)
_cpu = None


def cpu_lib():
    global _cpu
    if _cpu is None:
        _ensure_built()
        _cpu = C.CDLL(os.path.join(ORACLE_DIR, "libws_cpu_baseline.so"))
        _cpu.wscpu_update_tsdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _cpu.wscpu_register_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int]
    return _cpu


def cpu_update_tsdf(m: OracleMap, points, scanner_pos_vox, up, tau, max_weight, res, threads=1):
    pts = np.ascontiguousarray(points, dtype=np.int32)
    sp = np.ascontiguousarray(scanner_pos_vox, dtype=np.int32)
    u = np.ascontiguousarray(up, dtype=np.int32)
    return cpu_lib().wscpu_update_tsdf(_p(m.size), _p(m.pos), _p(m.offset), _p(m.data), _p(pts), pts.shape[0], _p(sp),
                                       _p(u), int(tau), int(max_weight), int(res), int(threads))


def cpu_register_cloud(m: OracleMap, points, T_in_rowmajor, max_iterations=200, it_weight_gradient=0.1, epsilon=0.03,
                       res=50, threads=0):
    pts = np.ascontiguousarray(points, dtype=np.int32).copy()
    T = colmajor(T_in_rowmajor)
    out = np.zeros(16, dtype=np.float32)
    it = cpu_lib().wscpu_register_cloud(_p(m.size), _p(m.pos), _p(m.offset), _p(m.data), _p(pts), pts.shape[0], _p(T),
                                        int(max_iterations), C.c_float(it_weight_gradient), C.c_float(epsilon),
                                        int(res), _p(out), int(threads))
    return out.reshape(4, 4).T.copy(), it, pts


# ---------------------------------------------------------------- oracle/_ref (reference headers, this container only)
def ref_lib():
    so = os.path.join(ORACLE_DIR, "_ref", "libws_ref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.ref_map_create.restype = C.c_void_p
    L.ref_map_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_map_destroy.argtypes = [C.c_void_p]
    L.ref_get_index.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.ref_get_index.restype = C.c_int32
    L.ref_in_bounds.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.ref_in_bounds_pos.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.ref_in_bounds_neg.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.ref_value_unchecked_raw.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.ref_value_unchecked_raw.restype = C.c_uint32
    L.ref_write_unchecked.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int16, C.c_int16]
    L.ref_pack.restype = C.c_uint32
    L.ref_pack.argtypes = [C.c_int16, C.c_int16]
    L.ref_entry_value.restype = C.c_int16
    L.ref_entry_value.argtypes = [C.c_uint32]
    L.ref_entry_weight.restype = C.c_int16
    L.ref_entry_weight.argtypes = [C.c_uint32]
    L.ref_l2norm_i.restype = C.c_int32
    L.ref_l2norm_i.argtypes = [C.c_int32] * 3
    L.ref_l2norm_l.restype = C.c_int64
    L.ref_l2norm_l.argtypes = [C.c_int64] * 3
    L.ref_cross_i.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_ray_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_integrate_entry.restype = C.c_uint32
    L.ref_integrate_entry.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    L.ref_jacobi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_matrix4_layout.argtypes = [C.c_void_p]
    L.ref_matrix6_layout.argtypes = [C.c_void_p]
    return L


def preprocess(xyz_f32, pose_rowmajor, res):
    """App::preprocess (app.cpp:119-148): (n, k>=3) float32 metres -> (m, 3) int32 mm, first-occurrence order."""
    L = lib()
    a = np.ascontiguousarray(xyz_f32, dtype=np.float32)
    n, stride = a.shape
    T = np.ascontiguousarray(np.asarray(pose_rowmajor, dtype=np.float32).reshape(4, 4).T).reshape(16)
    out = np.zeros((max(n, 1), 3), dtype=np.int32)
    m = L.wso_preprocess(a.ctypes.data_as(C.c_void_p), n, stride, T.ctypes.data_as(C.c_void_p), int(res), out.ctypes.data_as(C.c_void_p))
    return out[:m].copy()
