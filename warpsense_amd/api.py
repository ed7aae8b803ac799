"""Host-side mirror of the reference's device API over the C ABI (include/warpsense_hip.h).

Same names, argument meaning and error behaviour as the reference classes, so tests read like the
reference's own (test/cuda.cpp, test/pcd_registration.cpp):

    DeviceMap            include/warpsense/cuda/device_map.h:32-164     (non-owning host view)
    DeviceMapMemWrapper  include/warpsense/cuda/device_map_wrapper.h:10-36
    TSDFCuda             include/warpsense/cuda/update_tsdf.h:9-34
    RegistrationCuda     include/warpsense/cuda/registration.h:10-45
    TSDFMapping          src/warpsense/tsdf_mapping.cpp:30-95            (ROS-free constructor path)
    TSDFRegistration     src/warpsense/tsdf_registration.cpp:22-96
    pause / cleanup      src/warpsense/cuda/cleanup.cu

Matrices handed to these classes are ordinary numpy 4x4 arrays (math layout, M[i, j]); the column-major
flattening the C ABI wants (rmagine::Matrix4x4f / Eigen) happens here.  All compute runs in the HIP
library; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import (WS_INTEGRATE_DENSE, WS_INTEGRATE_SPARSE, WS_MAP_AVG, WS_MAP_NEW, WS_REG_ALL_POINTS,
                   WS_REG_COMPAT_REFERENCE_LAUNCH, WsError, check)

MATRIX_RESOLUTION = 32768  # include/warpsense/consts.h:12-13
WEIGHT_RESOLUTION = 64     # include/warpsense/consts.h:9-10


def _ptr(a):
    """void* of a numpy array, a ctypes array or a torch tensor (host or device)."""
    if a is None:
        return None
    if isinstance(a, C.Array):
        return a
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _is_device(a) -> bool:
    return hasattr(a, "is_cuda") and bool(a.is_cuda)


_I3 = C.c_int32 * 3


def _i3(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.int32).reshape(3))


def _i3c(v):
    """three int32 for the C ABI only; tuples and lists skip numpy (this sits between two scans of a stream)"""
    if isinstance(v, (tuple, list)) and len(v) == 3:
        return _I3(int(v[0]), int(v[1]), int(v[2]))
    return _i3(v)


def _colmajor(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T).reshape(16)


def pack_entry(value, weight):
    """TSDFEntry raw word: low 16 bits value, high 16 bits weight (include/map/tsdf.h:16-23)."""
    v = np.asarray(value).astype(np.int16).astype(np.uint16).astype(np.uint32)
    w = np.asarray(weight).astype(np.int16).astype(np.uint16).astype(np.uint32)
    return v | (w << np.uint32(16))


def unpack_entry(raw):
    raw = np.asarray(raw, dtype=np.uint32)
    return (raw & 0xFFFF).astype(np.uint16).astype(np.int16), (raw >> 16).astype(np.uint16).astype(np.int16)


class Context:
    """One per process and GPU: device + HIP stream (the reference uses the implicit CUDA context)."""

    _default = None
    _lock = threading.Lock()

    def __init__(self, device_id: int = -1):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.ws_ctx_create(int(device_id), C.byref(h)), "ws_ctx_create")
        self.handle = h

    @classmethod
    def default(cls) -> "Context":
        with cls._lock:
            if cls._default is None:
                cls._default = cls()
            return cls._default

    def set_stream(self, hip_stream_ptr):
        check(self._L.ws_ctx_set_stream(self.handle, C.c_void_p(hip_stream_ptr) if hip_stream_ptr else None),
              "ws_ctx_set_stream")

    def use_torch_stream(self):
        """Run the library's work on torch's CURRENT stream, so that torch operations (copies of the 44 sums, RCCL
        collectives) and the library's kernels are ordered by the stream alone.  The C ABI takes a hipStream_t and reads
        NULL as "the context's own stream", so torch's legacy default stream (handle 0) cannot be named there: in that case
        a new stream is created and made torch's current stream (for this thread) first."""
        import torch
        cur = torch.cuda.current_stream()
        if cur.cuda_stream == 0:
            # SIDE EFFECT, for the calling thread: torch's current stream becomes a new stream (restored by
            # restore_torch_stream() / close()).  The new stream first waits for everything already queued on the default
            # stream -- uploads or preprocessing the caller enqueued there -- so nothing queued earlier can race with later work.
            self._torch_prev_stream = cur
            new = torch.cuda.Stream()
            new.wait_stream(cur)
            torch.cuda.set_stream(new)
            cur = new
        self._torch_stream = cur  # keep it alive
        self.set_stream(cur.cuda_stream)

    def restore_torch_stream(self):
        """Undo use_torch_stream()'s stream switch: the library goes back to its own stream (after the work queued so far)
        and torch's current stream for this thread is the one it was before."""
        prev = getattr(self, "_torch_prev_stream", None)
        if prev is None:
            return
        import torch
        cur = getattr(self, "_torch_stream", None)
        if self.handle:
            self.set_stream(None)  # synchronises the stream it leaves
        if cur is not None:
            prev.wait_stream(cur)
        torch.cuda.set_stream(prev)
        self._torch_prev_stream = None
        self._torch_stream = None

    def sync(self):
        check(self._L.ws_sync(self.handle), "ws_sync")

    # hipEvent timing of kernel classes (bench.py)
    def prof_enable(self, mask: int):
        check(self._L.ws_prof_enable(self.handle, int(mask)), "ws_prof_enable")

    def prof_reset(self):
        check(self._L.ws_prof_reset(self.handle), "ws_prof_reset")

    def prof_read(self, cls: int):
        ms, n = C.c_double(0), C.c_int64(0)
        check(self._L.ws_prof_read(self.handle, int(cls), C.byref(ms), C.byref(n)), "ws_prof_read")
        return ms.value, n.value

    def close(self):
        if self.handle:
            try:
                self.restore_torch_stream()
            except Exception:  # noqa: BLE001 - closing must not fail on a torch that is already shutting down
                pass
            self._L.ws_ctx_destroy(self.handle)
            self.handle = None


def pause(ctx: Context | None = None):
    """cuda::pause() — cleanup.cu:3-6."""
    (ctx or Context.default()).sync()


def cleanup():
    """cuda::cleanup() — cleanup.cu:8-11.  Resets the device: every handle becomes invalid."""
    check(_lib.load().ws_device_reset(), "ws_device_reset")
    Context._default = None


class DeviceMap:
    """Non-owning view of a ring-buffer local map in HOST memory: size, offset, data, pos
    (include/warpsense/cuda/device_map.h:32-48).  The arrays are shared with whoever owns the map."""

    def __init__(self, size, offset, data, pos):
        self.size_ = np.asarray(size, dtype=np.int32).reshape(3)
        self.offset_ = np.asarray(offset, dtype=np.int32).reshape(3)
        self.pos_ = np.asarray(pos, dtype=np.int32).reshape(3)
        self.data_ = data
        if data is not None:
            assert data.dtype == np.uint32 and data.flags["C_CONTIGUOUS"]
            assert data.size == int(np.prod(self.size_.astype(np.int64)))

    def n_voxels(self) -> int:
        return int(np.prod(self.size_.astype(np.int64)))


def pose_to_values(pose, scale: float) -> np.ndarray:
    """The 7 floats HDF5GlobalMap::write_pose stores (hdf5_global_map.cpp:187-197): translation / scale and the
    rotation as a quaternion (x, y, z, w), each rounded to 3 decimals.  The quaternion follows Eigen's
    Quaternionf(Matrix3f) (trace branch / largest diagonal branch) in float32."""
    P = np.asarray(pose, dtype=np.float64).reshape(4, 4)
    out = np.zeros(7, dtype=np.float32)
    for k in range(3):
        # (double / float) * 1000.0f, std::round (half away from zero), / 1000.0f, stored as float
        v = P[k, 3] / float(np.float32(scale)) * 1000.0
        out[k] = np.float32(np.trunc(v + np.copysign(0.5, v)) / 1000.0)
    m = P[:3, :3].astype(np.float32)
    f = np.float32
    q = np.zeros(4, dtype=np.float32)  # x y z w
    t = f(m[0, 0] + m[1, 1] + m[2, 2])
    if t > 0:
        t = f(np.sqrt(f(t + f(1))))
        q[3] = f(f(0.5) * t)
        t = f(f(0.5) / t)
        q[0] = f(f(m[2, 1] - m[1, 2]) * t)
        q[1] = f(f(m[0, 2] - m[2, 0]) * t)
        q[2] = f(f(m[1, 0] - m[0, 1]) * t)
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = f(np.sqrt(f(f(f(m[i, i] - m[j, j]) - m[k, k]) + f(1))))
        q[i] = f(f(0.5) * t)
        t = f(f(0.5) / t)
        q[3] = f(f(m[k, j] - m[j, k]) * t)
        q[j] = f(f(m[j, i] + m[i, j]) * t)
        q[k] = f(f(m[k, i] + m[i, k]) * t)
    for n in range(4):
        v = f(q[n] * f(1000.0))
        out[3 + n] = f(f(np.trunc(v + np.copysign(f(0.5), v))) / f(1000.0))
    return out


class GlobalMap:
    """HDF5GlobalMap (src/map/hdf5_global_map.cpp): the world is cut into 64^3-voxel chunks of raw uint32
    entries, index x*4096 + y*64 + z inside a chunk (:53-57); chunks never seen hold the default entry.

    Without `filename` every chunk stays in memory.  With `filename` the map is the reference's .h5 file
    (layout in include/warpsense_h5.h, written through libwarpsense_h5.so): at most NUM_CHUNKS = 64 chunks are
    active, the least recently used one is written to the file when a 65th is needed (:59-135), write_back()
    flushes the active ones (:160-176), write_pose()/write_meta() as :178-221."""

    CHUNK_SIZE = 64
    NUM_CHUNKS = 64  # hdf5_global_map.h:78

    def __init__(self, default_value, default_weight=0, filename: str | None = None, map_params=None, open_existing: bool = False):
        from collections import OrderedDict
        self.default_raw = int(pack_entry(default_value, default_weight))
        self.lock = threading.RLock()  # the map-shift worker saves slabs while the scan thread writes poses / loads boxes
        self.chunks: "OrderedDict[tuple[int, int, int], np.ndarray]" = OrderedDict()
        self._file = None
        self._filename = filename
        if filename is not None:
            self._H = _lib.load_h5()
            h = C.c_void_p()
            if open_existing:
                _lib.check_h5(self._H.ws_h5_open(filename.encode(), 1, C.byref(h)), "ws_h5_open")
            else:
                _lib.check_h5(self._H.ws_h5_create(filename.encode(), C.byref(h)), "ws_h5_create")
            self._file = h
            # which chunks the file holds, without asking HDF5 again (the map shift looks this up per entering chunk)
            self._in_file: set[tuple[int, int, int]] = set()
            if open_existing:
                n = C.c_int64(0)
                _lib.check_h5(self._H.ws_h5_num_chunks(h, C.byref(n)), "ws_h5_num_chunks")
                if n.value:
                    pos = np.zeros((n.value, 3), dtype=np.int32)
                    _lib.check_h5(self._H.ws_h5_list_chunks(h, _ptr(pos), n.value, C.byref(n)), "ws_h5_list_chunks")
                    self._in_file = {tuple(int(v) for v in p) for p in pos[:n.value]}
            if map_params is not None and not open_existing:
                self.write_meta(map_params)

    def filename(self):
        return self._filename

    # -- chunk cache --------------------------------------------------------------------------------
    def _write_chunk(self, key, data):
        _lib.check_h5(self._H.ws_h5_write_chunk(self._file, key[0], key[1], key[2], _ptr(data)), "ws_h5_write_chunk")
        self._in_file.add(key)

    def has_chunk(self, cx, cy, cz) -> bool:
        """True if the map holds data for this chunk (in memory or in the file) — a chunk never seen is all default."""
        key = (int(cx), int(cy), int(cz))
        with self.lock:
            return key in self.chunks or (self._file is not None and key in self._in_file)

    def activate_chunk(self, cx, cy, cz) -> np.ndarray:
        key = (int(cx), int(cy), int(cz))
        c = self.chunks.get(key)
        if c is not None:
            if self._file is not None:
                self.chunks.move_to_end(key)  # age 0
            return c
        c = np.empty(self.CHUNK_SIZE ** 3, dtype=np.uint32)
        found = C.c_int32(0)
        if self._file is not None:
            _lib.check_h5(self._H.ws_h5_read_chunk(self._file, key[0], key[1], key[2], _ptr(c), C.byref(found)), "ws_h5_read_chunk")
        if not found.value:
            c.fill(self.default_raw)
        if self._file is not None and len(self.chunks) >= self.NUM_CHUNKS:
            old_key, old = self.chunks.popitem(last=False)  # the oldest chunk goes to the file
            self._write_chunk(old_key, old)
        self.chunks[key] = c
        return c

    def get_value(self, x, y, z):
        cs = self.CHUNK_SIZE
        cx, cy, cz = int(x) // cs, int(y) // cs, int(z) // cs  # floor_divide, also for negative coordinates
        c = self.activate_chunk(cx, cy, cz)
        v, w = unpack_entry(c[(int(x) - cx * cs) * cs * cs + (int(y) - cy * cs) * cs + (int(z) - cz * cs)])
        return int(v), int(w)

    def set_value(self, x, y, z, value, weight):
        cs = self.CHUNK_SIZE
        cx, cy, cz = int(x) // cs, int(y) // cs, int(z) // cs
        c = self.activate_chunk(cx, cy, cz)
        c[(int(x) - cx * cs) * cs * cs + (int(y) - cy * cs) * cs + (int(z) - cz * cs)] = pack_entry(value, weight)

    def write_back(self):
        if self._file is None:
            return
        with self.lock:
            for key, data in self.chunks.items():
                self._write_chunk(key, data)
            _lib.check_h5(self._H.ws_h5_flush(self._file), "ws_h5_flush")

    def write_pose(self, pose, scale: float):
        if self._file is None:
            raise WsError("write_pose: this GlobalMap has no file")
        vals = pose_to_values(pose, scale)
        with self.lock:  # the map-shift worker may be writing chunks: the HDF5 library is used by one thread at a time
            _lib.check_h5(self._H.ws_h5_write_pose(self._file, _ptr(vals)), "ws_h5_write_pose")
        return vals

    def write_meta(self, p):
        """p: MapParams (tau, size in voxels, max_distance, resolution, max_weight scaled like map_params.h:88-90)."""
        size = np.asarray(p.size_voxels(), dtype=np.int32)
        _lib.check_h5(self._H.ws_h5_write_meta(self._file, int(p.tau), _ptr(size), C.c_float(p.max_distance), int(p.resolution),
                                               int(p.max_weight)), "ws_h5_write_meta")

    def close(self):
        if self._file is not None:
            self.write_back()
            self._H.ws_h5_close(self._file)
            self._file = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _box(self, lo, hi, buf, save: bool):
        """Move a dense world-voxel box (inclusive, x major / z fastest) between `buf` and the chunks."""
        cs = self.CHUNK_SIZE
        lo, hi = np.asarray(lo, dtype=np.int64), np.asarray(hi, dtype=np.int64)
        ext = hi - lo + 1
        box = buf.reshape(int(ext[0]), int(ext[1]), int(ext[2]))
        c0, c1 = np.floor_divide(lo, cs), np.floor_divide(hi, cs)
        for cx in range(c0[0], c1[0] + 1):
            for cy in range(c0[1], c1[1] + 1):
                for cz in range(c0[2], c1[2] + 1):
                    base = np.array([cx, cy, cz], dtype=np.int64) * cs
                    a = np.maximum(lo, base)
                    b = np.minimum(hi, base + cs - 1)
                    sl_c = tuple(slice(int(a[k] - base[k]), int(b[k] - base[k]) + 1) for k in range(3))
                    sl_b = tuple(slice(int(a[k] - lo[k]), int(b[k] - lo[k]) + 1) for k in range(3))
                    with self.lock:  # per chunk: a pose written by the scan thread never waits for a whole slab
                        chunk = self.activate_chunk(cx, cy, cz).reshape(cs, cs, cs)
                        if save:
                            chunk[sl_c] = box[sl_b]
                        else:
                            box[sl_b] = chunk[sl_c]

    def save_box(self, lo, hi, buf):
        self._box(lo, hi, buf, True)

    def load_box(self, lo, hi) -> np.ndarray:
        ext = np.asarray(hi, dtype=np.int64) - np.asarray(lo, dtype=np.int64) + 1
        buf = np.empty(int(np.prod(ext)), dtype=np.uint32)
        self._box(lo, hi, buf, False)
        return buf


class LocalMap:
    """HDF5LocalMap without the file (src/map/hdf5_local_map.cpp): a 3-D ring buffer of TSDF entries around `pos`.
    Sizes are forced odd, offset = size/2, every voxel starts as the global map's default entry (:5-20);
    shift() moves the window, saving the slabs that leave to the global map and loading the ones that enter
    (:53-118).  Owns the host arrays a DeviceMap views."""

    def __init__(self, sx, sy, sz, default_value, default_weight=0, global_map: GlobalMap | None = None, host_voxels: bool = True):
        self.size = np.array([s if s % 2 == 1 else s + 1 for s in (int(sx), int(sy), int(sz))], dtype=np.int32)
        self.pos = np.zeros(3, dtype=np.int32)
        self.offset = (self.size // 2).astype(np.int32)
        self.map_ = global_map or GlobalMap(default_value, default_weight)
        # host_voxels=False: the window lives on the device only (a 2049^3 window is 34 GB; the device-side shift and
        # export never need the host copy)
        self.data = np.full(int(np.prod(self.size.astype(np.int64))), self.map_.default_raw, dtype=np.uint32) if host_voxels else None

    def device_map(self) -> DeviceMap:
        return DeviceMap(self.size, self.offset, self.data, self.pos)

    def get_index(self, x, y, z) -> int:
        s, p, o = self.size.astype(np.int64), self.pos.astype(np.int64), self.offset.astype(np.int64)
        xi = (x - p[0] + o[0] + s[0]) % s[0]
        yi = (y - p[1] + o[1] + s[1]) % s[1]
        zi = (z - p[2] + o[2] + s[2]) % s[2]
        return int((xi * s[1] + yi) * s[2] + zi)

    def in_bounds(self, x, y, z) -> bool:
        d = np.abs(np.array([x, y, z], dtype=np.int64) - self.pos)
        return bool(np.all(d <= self.size // 2))

    def value(self, x, y, z):
        if not self.in_bounds(x, y, z):
            raise IndexError(f"Index out of bounds: {x}; {y}; {z}")  # std::out_of_range, hdf5_local_map.h:172-181
        v, w = unpack_entry(self.data[self.get_index(x, y, z)])
        return int(v), int(w)

    def set_value(self, x, y, z, value, weight):
        if not self.in_bounds(x, y, z):
            raise IndexError(f"Index out of bounds: {x}; {y}; {z}")
        self.data[self.get_index(x, y, z)] = pack_entry(value, weight)

    # -- save_load_area<save> (hdf5_local_map.cpp:120-198): one pass per 64^3 chunk the box touches
    def _area(self, start, end, save: bool):
        cs = GlobalMap.CHUNK_SIZE
        start, end = np.minimum(start, end).astype(np.int64), np.maximum(start, end).astype(np.int64)
        s, p, o = self.size.astype(np.int64), self.pos.astype(np.int64), self.offset.astype(np.int64)
        c0, c1 = np.floor_divide(start, cs), np.floor_divide(end, cs)
        for cx in range(c0[0], c1[0] + 1):
            for cy in range(c0[1], c1[1] + 1):
                for cz in range(c0[2], c1[2] + 1):
                    chunk = self.map_.activate_chunk(cx, cy, cz)
                    base = np.array([cx, cy, cz], dtype=np.int64) * cs
                    lo = np.maximum(start, base) - base
                    hi = np.minimum(end, base + cs - 1) - base
                    d = [np.arange(lo[k], hi[k] + 1, dtype=np.int64) for k in range(3)]
                    g = [d[k] + base[k] for k in range(3)]
                    r = [(g[k] - p[k] + o[k] + s[k]) % s[k] for k in range(3)]
                    ring = ((r[0][:, None, None] * s[1] + r[1][None, :, None]) * s[2] + r[2][None, None, :]).reshape(-1)
                    loc = (d[0][:, None, None] * cs * cs + d[1][None, :, None] * cs + d[2][None, None, :]).reshape(-1)
                    if save:
                        chunk[loc] = self.data[ring]
                    else:
                        self.data[ring] = chunk[loc]

    def shift(self, new_pos):
        """HDF5LocalMap::shift (hdf5_local_map.cpp:53-118), axis by axis."""
        new_pos = np.asarray(new_pos, dtype=np.int64)
        diff = new_pos - self.pos
        assert np.all(np.abs(diff) <= self.size)
        for axis in range(3):
            if diff[axis] == 0:
                continue
            start = self.pos.astype(np.int64) - self.size // 2
            end = self.pos.astype(np.int64) + self.size // 2
            if diff[axis] > 0:
                end[axis] = start[axis] + diff[axis] - 1
            else:
                start[axis] = end[axis] + diff[axis] + 1
            self._area(start, end, save=True)
            self.pos[axis] += diff[axis]
            self.offset[axis] = (self.offset[axis] + diff[axis] + self.size[axis]) % self.size[axis]
            start = self.pos.astype(np.int64) - self.size // 2
            end = self.pos.astype(np.int64) + self.size // 2
            if diff[axis] > 0:
                start[axis] = end[axis] - (diff[axis] - 1)
            else:
                end[axis] = start[axis] - diff[axis] - 1
            self._area(start, end, save=False)

    def write_back(self):
        """hdf5_local_map.cpp:210-217: save the whole window to the global map."""
        self._area(self.pos.astype(np.int64) - self.size // 2, self.pos.astype(np.int64) + self.size // 2, save=True)


class DeviceMapMemWrapper:
    """Device copy of one map (device_map_wrapper.h:10-36); owned by a TSDFCuda."""

    def __init__(self, tsdf: "TSDFCuda", which: int):
        self._t = tsdf
        self._which = which

    def to_device(self, existing_map: DeviceMap):
        t = self._t
        check(t._L.ws_map_upload(t.handle, self._which, _ptr(_i3(existing_map.size_)), _ptr(_i3(existing_map.pos_)),
                                 _ptr(_i3(existing_map.offset_)), _ptr(existing_map.data_)), "ws_map_upload")

    def update_params(self, existing_map: DeviceMap):
        t = self._t
        check(t._L.ws_map_set_params(t.handle, self._which, _ptr(_i3(existing_map.size_)),
                                     _ptr(_i3(existing_map.pos_)), _ptr(_i3(existing_map.offset_))), "ws_map_set_params")

    def to_host(self, existing_map: DeviceMap):
        t = self._t
        size, pos, off = np.zeros(3, np.int32), np.zeros(3, np.int32), np.zeros(3, np.int32)
        check(t._L.ws_map_download(t.handle, self._which, _ptr(size), _ptr(pos), _ptr(off), _ptr(existing_map.data_)),
              "ws_map_download")
        existing_map.size_[:] = size
        existing_map.pos_[:] = pos
        existing_map.offset_[:] = off

    def extract_box(self, lo, hi) -> np.ndarray:
        """Voxels of the inclusive world-voxel box [lo, hi] (x major, z fastest) — ws_map_extract_box."""
        t = self._t
        lo, hi = _i3(lo), _i3(hi)
        out = np.empty(int(np.prod((hi - lo + 1).astype(np.int64))), dtype=np.uint32)
        check(t._L.ws_map_extract_box(t.handle, self._which, _ptr(lo), _ptr(hi), _ptr(out)), "ws_map_extract_box")
        return out

    def insert_box(self, lo, hi, data):
        t = self._t
        lo, hi = _i3(lo), _i3(hi)
        data = np.ascontiguousarray(data, dtype=np.uint32)
        assert data.size == int(np.prod((hi - lo + 1).astype(np.int64)))
        check(t._L.ws_map_insert_box(t.handle, self._which, _ptr(lo), _ptr(hi), _ptr(data)), "ws_map_insert_box")

    def dev(self):
        return self._t.handle

    def device_ptr(self) -> int:
        return int(self._t._L.ws_map_device_data(self._t.handle, self._which) or 0)


class TSDFCuda:
    """cuda::TSDFCuda (update_tsdf.h:9-34, update_tsdf.cu:130-221)."""

    n_max_points_ = 1_000_000

    def __init__(self, existing_map: DeviceMap, tau: int, max_weight: int, map_resolution: int, ctx: Context | None = None):
        self.ctx = ctx or Context.default()
        self._L = self.ctx._L
        self.tau_, self.max_weight_, self.map_resolution_ = int(tau), int(max_weight), int(map_resolution)
        self.n_voxels_ = existing_map.n_voxels()
        h = C.c_void_p()
        check(self._L.ws_map_create(self.ctx.handle, _ptr(_i3(existing_map.size_)), _ptr(_i3(existing_map.pos_)),
                                    _ptr(_i3(existing_map.offset_)), _ptr(existing_map.data_), self.tau_, self.max_weight_,
                                    self.map_resolution_, C.byref(h)), "ws_map_create")
        self.handle = h
        self._avg = DeviceMapMemWrapper(self, WS_MAP_AVG)
        self._new = DeviceMapMemWrapper(self, WS_MAP_NEW)
        self._scan_in_flight = None  # the device tensor of the last update (kept until the next one is enqueued behind it)

    # -- the three update_tsdf overloads of the reference (update_tsdf.cu:143-191)
    def update_tsdf(self, scan_points, scanner_pos, up, result: DeviceMap | None = None, latest_map: DeviceMap | None = None):
        n = int(scan_points.shape[0])
        sp, u = _i3c(scanner_pos), _i3c(up)
        if _is_device(scan_points):
            # ws_tsdf_update_dev returns after its launches: the tensor must not go back to torch's allocator before the kernels have
            # read it (torch's allocator only knows torch's streams) -- keep it until the next update has been enqueued behind it
            prev = self._scan_in_flight
            rc = self._L.ws_tsdf_update_dev(self.handle, _ptr(scan_points), n, _ptr(sp), _ptr(u))
            self._scan_in_flight = scan_points
            del prev
        else:
            pts = np.ascontiguousarray(scan_points, dtype=np.int32)
            rc = self._L.ws_tsdf_update(self.handle, _ptr(pts), n, _ptr(sp), _ptr(u))
        if rc == -3:
            # update_tsdf.cu:146-150: message on stderr, no work, no exception
            import sys
            print("HIP Error: " + self._L.ws_last_error().decode(), file=sys.stderr)
            return
        check(rc, "ws_tsdf_update")
        if result is not None:
            self._avg.to_host(result)
        if latest_map is not None:
            self._new.to_host(latest_map)

    def scatter(self, scan_points_dev, scanner_pos, up):
        """cu_min_tsdf_krnl alone (parity tests): leaves the resolved scan in new_map."""
        prev = self._scan_in_flight  # (see update_tsdf)
        check(self._L.ws_tsdf_scatter_dev(self.handle, _ptr(scan_points_dev), int(scan_points_dev.shape[0]),
                                          _ptr(_i3c(scanner_pos)), _ptr(_i3c(up))), "ws_tsdf_scatter_dev")
        self._scan_in_flight = scan_points_dev
        del prev

    def integrate(self):
        check(self._L.ws_tsdf_integrate(self.handle), "ws_tsdf_integrate")

    def set_integrate(self, mode: int):
        check(self._L.ws_tsdf_set_integrate(self.handle, int(mode)), "ws_tsdf_set_integrate")

    def set_capacity(self, records: int):
        check(self._L.ws_tsdf_set_capacity(self.handle, int(records)), "ws_tsdf_set_capacity")

    def debug_chunk_policy(self, budget_bytes: int, est_shift: int):
        """test entry: size the chunk buffer by estimate (see ws_debug_tsdf_chunk_policy)"""
        check(self._L.ws_debug_tsdf_chunk_policy(self.handle, int(budget_bytes), int(est_shift)), "ws_debug_tsdf_chunk_policy")

    def stats(self, raise_on_error: bool = False) -> dict:
        st = _lib.TsdfStats()
        rc = self._L.ws_tsdf_stats(self.handle, C.byref(st))
        out = {"contested_voxels": st.contested_voxels, "records": st.records, "tiles": st.tiles, "error_flags": st.error_flags,
               "runs": st.runs, "free_space_hits": st.free_space_hits, "record_slots": st.record_slots,
               "record_capacity": st.record_capacity, "hash_entries": st.hash_entries, "status": rc}
        if rc != 0 and raise_on_error:
            check(rc, "ws_tsdf_stats")
        return out

    def device_map(self):
        return self.handle

    def avg_map(self) -> DeviceMapMemWrapper:
        return self._avg

    def new_map(self) -> DeviceMapMemWrapper:
        return self._new

    def close(self):
        if self.handle:
            self._L.ws_map_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RegistrationCuda:
    """cuda::RegistrationCuda (registration.h:10-45, registration.cu:259-368)."""

    def __init__(self, map_: DeviceMap | None = None, ctx: Context | None = None, flags: int = WS_REG_ALL_POINTS):
        self.ctx = ctx or Context.default()
        self._L = self.ctx._L
        self.flags = int(flags)
        h = C.c_void_p()
        check(self._L.ws_reg_create(self.ctx.handle, 128 * 1024, C.byref(h)), "ws_reg_create")
        self.handle = h
        self.curr_n_points = 0

    def prepare_registration(self, points):
        n = int(points.shape[0])
        self.curr_n_points = n
        if _is_device(points):
            # (an asynchronous device-to-device copy on the context's stream: like TSDFCuda.update_tsdf, keep the tensor away from
            # torch's allocator until the next cloud's copy has been enqueued behind it)
            prev = getattr(self, "_cloud_in_flight", None)
            check(self._L.ws_reg_prepare_dev(self.handle, _ptr(points), n), "ws_reg_prepare_dev")
            self._cloud_in_flight = points
            del prev
        else:
            pts = np.ascontiguousarray(points, dtype=np.int32)
            check(self._L.ws_reg_prepare(self.handle, _ptr(pts), n), "ws_reg_prepare")

    def perform_registration(self, map_dev, pretransform, map_resolution: int):
        """Returns (h 6x6 int64 math layout, g[6] int64, e, c) — the out-parameters of the reference."""
        T = _colmajor(pretransform)
        h = np.zeros(36, dtype=np.int64)
        g = np.zeros(6, dtype=np.int64)
        e, c = C.c_int32(0), C.c_int32(0)
        check(self._L.ws_reg_iterate(self.handle, map_dev, _ptr(T), int(map_resolution), self.flags, _ptr(h), _ptr(g),
                                     C.byref(e), C.byref(c)), "ws_reg_iterate")
        return h.reshape(6, 6).T.copy(), g, e.value, c.value

    def register_cloud(self, map_dev, pretransform, max_iterations, it_weight_gradient, epsilon, map_resolution):
        # (two ctypes buffers kept for the life of the object: numpy's .ctypes costs 3 us per array, and this call sits in the gap
        # between the end of one registration and the first kernel of the next update)
        buf = getattr(self, "_rc_buf", None)
        if buf is None:
            buf = self._rc_buf = ((C.c_float * 16)(), (C.c_float * 16)(), C.c_int32(0))
        T_c, out_c, it = buf
        T_c[:] = np.asarray(pretransform, dtype=np.float32).reshape(4, 4).T.reshape(16).tolist()
        check(self._L.ws_register_cloud(self.handle, map_dev, T_c, int(max_iterations), C.c_float(it_weight_gradient),
                                        C.c_float(epsilon), int(map_resolution), self.flags, out_c, C.byref(it)),
              "ws_register_cloud")
        return np.array(out_c, dtype=np.float32).reshape(4, 4).T.copy(), it.value

    def last_sums(self):
        """(h 6x6 int64, g[6], e, c) the last Gauss-Newton update of the last register_cloud was made from (test entry)"""
        out = np.empty(44, dtype=np.int64)
        check(self._L.ws_debug_reg_sums(self.handle, _ptr(out)), "ws_debug_reg_sums")
        return out[:36].reshape(6, 6).T.copy(), out[36:42].copy(), int(out[42]), int(out[43])

    def set_loop(self, mode: int):
        """WS_REG_LOOP_RESIDENT (one launch, grid barrier; default) or WS_REG_LOOP_LAUNCHES (one launch per iteration)."""
        check(self._L.ws_reg_set_loop(self.handle, int(mode)), "ws_reg_set_loop")

    def close(self):
        if self.handle:
            self._L.ws_reg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePoints:
    """(n, 3) int32 points resident on the device (what ScanPreprocessor returns); accepted wherever a CUDA
    tensor of points is (TSDFCuda.update_tsdf, RegistrationCuda.prepare_registration)."""

    is_cuda = True

    def __init__(self, ptr: int, n: int, owner):
        self._ptr, self.shape, self._owner = int(ptr), (int(n), 3), owner

    def data_ptr(self) -> int:
        return self._ptr

    def __len__(self):
        return self.shape[0]

    def to_host(self) -> np.ndarray:
        return self._owner.download()


class ScanPreprocessor:
    """App::preprocess on the device (src/warpsense/app.cpp:119-148): sensor cloud in float metres -> distinct
    voxel-centre points in int mm, transformed by the pose, in first-occurrence order."""

    def __init__(self, max_points: int = 128 * 1024, ctx: Context | None = None):
        self.ctx = ctx or Context.default()
        self._L = self.ctx._L
        h = C.c_void_p()
        check(self._L.ws_scan_create(self.ctx.handle, int(max_points), C.byref(h)), "ws_scan_create")
        self.handle = h
        self.n_out = 0

    def preprocess(self, cloud, pose, map_resolution: int) -> DevicePoints:
        """cloud: (n, k >= 3) float32, numpy or CUDA tensor (x y z first); pose: 4x4, translation in mm."""
        n, stride = int(cloud.shape[0]), int(cloud.shape[1])
        T = _colmajor(pose)
        out = C.c_size_t(0)
        if _is_device(cloud):
            check(self._L.ws_scan_preprocess_dev(self.handle, _ptr(cloud), n, stride, _ptr(T), int(map_resolution), C.byref(out)),
                  "ws_scan_preprocess_dev")
        else:
            a = np.ascontiguousarray(cloud, dtype=np.float32)
            check(self._L.ws_scan_preprocess(self.handle, _ptr(a), n, stride, _ptr(T), int(map_resolution), C.byref(out)),
                  "ws_scan_preprocess")
        self.n_out = int(out.value)
        return DevicePoints(self._L.ws_scan_points_dev(self.handle) or 0, self.n_out, self)

    def download(self) -> np.ndarray:
        pts = np.zeros((max(self.n_out, 1), 3), dtype=np.int32)
        n = C.c_size_t(0)
        check(self._L.ws_scan_download(self.handle, _ptr(pts), pts.shape[0], C.byref(n)), "ws_scan_download")
        return pts[:int(n.value)].copy()

    def close(self):
        if self.handle:
            self._L.ws_scan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def to_int_mat(mat):
    """(mat * MATRIX_RESOLUTION).cast<int>() — include/util/util.h:8-11."""
    return (np.asarray(mat, dtype=np.float32) * np.float32(MATRIX_RESOLUTION)).astype(np.int32)


def to_map(pose, map_resolution: int):
    """floor(translation / resolution) in float — include/util/util.h:52-56."""
    t = np.asarray(pose, dtype=np.float32)[:3, 3]
    return np.floor(t / np.float32(map_resolution)).astype(np.int32)


class MapParams:
    """The hot-path knobs of params/params.yaml with the scaling rules of include/params/map_params.h:100-114."""

    def __init__(self, resolution=64, max_distance=1.0, max_weight=10, size=(40.0, 40.0, 25.0), shift=10.0, initial_weight=0):
        self.resolution = int(resolution)
        self.max_distance = float(max_distance)
        self.tau = int(max_distance * 1000)
        self.max_weight = int(max_weight * WEIGHT_RESOLUTION)
        self.size = tuple(int(s * 1000 / self.resolution) for s in size)
        self.shift = float(shift)
        self.initial_weight = int(initial_weight)

    def size_voxels(self):
        return self.size


class RegistrationParams:
    def __init__(self, max_iterations=200, it_weight_gradient=0.1, epsilon=0.03):
        self.max_iterations = int(max_iterations)
        self.it_weight_gradient = float(it_weight_gradient)
        self.epsilon = float(epsilon)


class Params:
    def __init__(self, map_params: MapParams | None = None, registration: RegistrationParams | None = None):
        self.map = map_params or MapParams()
        self.registration = registration or RegistrationParams()


class TSDFMapping:
    """cuda::TSDFMapping without ROS (the protected constructor path, tsdf_mapping.cpp:30-41)."""

    def __init__(self, params: Params, local_map: LocalMap, ctx: Context | None = None):
        self.params_ = params
        self.local_map_ = local_map
        self.cuda_map_ = local_map.device_map()
        self.tsdf_ = TSDFCuda(self.cuda_map_, params.map.tau, params.map.max_weight, params.map.resolution, ctx)
        self.mutex_ = threading.RLock()  # the reference's shared_mutex: one writer or many readers

    def convert_pose_to_gpu(self, pose):
        """tsdf_mapping.cpp:77-85: pos = floor(t/res) voxels, up = (R_int * (0,0,MR)) / MR."""
        # (R_int * (0, 0, MR) + 0) / MR == third column of R_int exactly (|R_int| <= 32768, no int overflow)
        up = to_int_mat(pose)[:3, 2].astype(np.int32)
        return to_map(pose, self.params_.map.resolution), up

    def update_tsdf(self, scan_points, pose=None, pos_rm=None, up_rm=None, result: DeviceMap | None = None):
        if pose is not None:
            pos_rm, up_rm = self.convert_pose_to_gpu(pose)
        with self.mutex_:
            self.tsdf_.update_tsdf(scan_points, pos_rm, up_rm, result=result)

    def tsdf(self) -> TSDFCuda:
        return self.tsdf_

    def shift_map(self, new_pos):
        """TSDFMapping::map_shift (tsdf_mapping.cpp:109-126) with the window moved ON THE DEVICE: per axis, the slab
        that leaves is packed and saved to the global map, pos/offset move, the slab that enters is loaded and
        unpacked (the three steps of HDF5LocalMap::shift, hdf5_local_map.cpp:53-118).  The reference copies the whole
        map device -> host -> device instead.  The host LocalMap only follows pos/offset; its voxel array is stale
        until avg_map().to_host() is called."""
        self.wait_shift()
        lm, avg, new = self.local_map_, self.tsdf_.avg_map(), self.tsdf_.new_map()
        new_pos = np.asarray(new_pos, dtype=np.int64)
        with self.mutex_:
            diff = new_pos - lm.pos
            assert np.all(np.abs(diff) <= lm.size)
            for axis in range(3):
                d = int(diff[axis])
                if d == 0:
                    continue
                half = lm.size.astype(np.int64) // 2
                start, end = lm.pos.astype(np.int64) - half, lm.pos.astype(np.int64) + half
                if d > 0:
                    end[axis] = start[axis] + d - 1
                else:
                    start[axis] = end[axis] + d + 1
                lm.map_.save_box(start, end, avg.extract_box(start, end))
                lm.pos[axis] += d
                lm.offset[axis] = (lm.offset[axis] + d + lm.size[axis]) % lm.size[axis]
                view = lm.device_map()
                avg.update_params(view)
                new.update_params(view)  # new_map is (tau, 0) everywhere: only its window moves
                start, end = lm.pos.astype(np.int64) - half, lm.pos.astype(np.int64) + half
                if d > 0:
                    start[axis] = end[axis] - (d - 1)
                else:
                    end[axis] = start[axis] - d - 1
                avg.insert_box(start, end, lm.map_.load_box(start, end))


    # ---- the same shift off the scan path ---------------------------------------------------------------
    def shift_map_async(self, new_pos):
        """TSDFMapping::map_shift as the reference runs it — on its own thread, the scans only wait for the swap
        (tsdf_mapping.cpp:97-136).  On the scan path: ws_shift_begin (device kernels: pack the leaving slabs into a staging
        buffer, move the window, fill the entering slabs with the default entry — stream-ordered, no waiting) and, only for
        revisited space, the upload of the chunks the global map already holds for the entering slabs.  The leaving slabs
        travel to the host on a second stream and a worker thread files them into the global map.  Same result as
        shift_map(); wait_shift() joins the worker (write_back() and the next shift do that themselves)."""
        self.wait_shift()  # one shift in flight; its slabs must be in the global map before entering data is looked up
        lm, avg = self.local_map_, self.tsdf_.avg_map()
        new_pos = _i3(new_pos)
        L = self.tsdf_._L
        ticket = C.c_void_p()
        n = 0
        self._shift_error = None

        def file_slabs():
            # the ticket is closed whatever happens, and a failure travels to wait_shift() (a daemon thread's traceback
            # would be the only trace of slabs that never reached the global map)
            try:
                file_slabs_body()
            except BaseException as exc:  # noqa: BLE001 - re-raised by wait_shift()
                self._shift_error = exc
            finally:
                L.ws_shift_end(ticket)

        def file_slabs_body():
            check(L.ws_shift_wait(ticket), "ws_shift_wait")
            slo, shi = np.zeros(3, dtype=np.int32), np.zeros(3, dtype=np.int32)
            elo, ehi = np.zeros(3, dtype=np.int32), np.zeros(3, dtype=np.int32)
            for i in range(n):
                data = C.c_void_p()
                check(L.ws_shift_slab(ticket, i, _ptr(slo), _ptr(shi), C.byref(data)), "ws_shift_slab")
                ext = shi.astype(np.int64) - slo + 1
                buf = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint32)), shape=(int(np.prod(ext)),))
                box = buf.reshape(int(ext[0]), int(ext[1]), int(ext[2]))
                # A corner that ENTERED with an earlier axis of this shift and leaves again with this one was packed as
                # default fill; the synchronous route would have loaded it from the global map and saved it back
                # unchanged.  Put the global map's own data there, so the save below changes nothing for it.
                for j in range(i):
                    check(L.ws_shift_entering(ticket, j, _ptr(elo), _ptr(ehi)), "ws_shift_entering")
                    a, b = np.maximum(slo, elo).astype(np.int64), np.minimum(shi, ehi).astype(np.int64)
                    if np.all(a <= b):
                        e = b - a + 1
                        sl = tuple(slice(int(a[k] - slo[k]), int(b[k] - slo[k]) + 1) for k in range(3))
                        box[sl] = lm.map_.load_box(a, b).reshape(int(e[0]), int(e[1]), int(e[2]))
                lm.map_.save_box(slo.copy(), shi.copy(), buf)

        with self.mutex_:
            check(L.ws_shift_begin(self.tsdf_.handle, _ptr(new_pos), int(lm.map_.default_raw), C.byref(ticket)), "ws_shift_begin")
            n = L.ws_shift_count(ticket)
            try:
                self._shift_enter(L, ticket, n, new_pos, lm, avg)
            except BaseException:
                # the window has moved on the device: the leaving slabs are still filed and the ticket is closed (an open
                # ticket makes every later ws_shift_begin fail); the failure itself goes to the caller
                file_slabs()
                raise
        self._shift_worker = threading.Thread(target=file_slabs, name="warpsense-map-shift", daemon=True)
        self._shift_worker.start()

    def _shift_enter(self, L, ticket, n, new_pos, lm, avg):
        """the host side of an asynchronous shift between ws_shift_begin and the start of the worker"""
        # the host view of the window follows (pos / offset per axis exactly like HDF5LocalMap::shift)
        diff = new_pos.astype(np.int64) - lm.pos
        for axis in range(3):
            d = int(diff[axis])
            lm.pos[axis] += d
            lm.offset[axis] = (lm.offset[axis] + d + lm.size[axis]) % lm.size[axis]
        # revisited space: chunks the global map already has overwrite the default fill
        cs = GlobalMap.CHUNK_SIZE
        lo, hi = np.zeros(3, dtype=np.int32), np.zeros(3, dtype=np.int32)
        for i in range(n):
            check(L.ws_shift_entering(ticket, i, _ptr(lo), _ptr(hi)), "ws_shift_entering")
            # a later axis step moves the window again: only the part of the slab still inside the FINAL window counts
            half = lm.size.astype(np.int64) // 2
            a = np.maximum(lo.astype(np.int64), lm.pos.astype(np.int64) - half)
            b = np.minimum(hi.astype(np.int64), lm.pos.astype(np.int64) + half)
            if np.any(a > b):
                continue
            c0, c1 = np.floor_divide(a, cs), np.floor_divide(b, cs)
            for cx in range(int(c0[0]), int(c1[0]) + 1):
                for cy in range(int(c0[1]), int(c1[1]) + 1):
                    for cz in range(int(c0[2]), int(c1[2]) + 1):
                        if not lm.map_.has_chunk(cx, cy, cz):
                            continue
                        base = np.array([cx, cy, cz], dtype=np.int64) * cs
                        sa, sb = np.maximum(a, base), np.minimum(b, base + cs - 1)
                        avg.insert_box(sa, sb, lm.map_.load_box(sa, sb))

    def reserve_shift(self, shift_voxels: int):
        """staging for asynchronous shifts of up to `shift_voxels` per axis (plus slack), allocated now instead of inside
        the first shift (a pinned allocation of that size takes tens of milliseconds)"""
        s = self.local_map_.size.astype(np.int64)
        d = int(shift_voxels) + 8
        total = int(d * (s[0] * s[1] + s[1] * s[2] + s[0] * s[2]))
        check(self.tsdf_._L.ws_shift_reserve(self.tsdf_.handle, total), "ws_shift_reserve")

    def wait_shift(self):
        w = getattr(self, "_shift_worker", None)
        if w is not None:
            w.join()
            self._shift_worker = None
            err, self._shift_error = getattr(self, "_shift_error", None), None
            if err is not None:
                raise WsError(f"asynchronous map shift: the leaving slabs were not filed into the global map ({err!r})") from err

    def write_back(self, box_lo=None, box_hi=None):
        """HDF5LocalMap::write_back + HDF5GlobalMap::write_back (hdf5_local_map.cpp:210-217, app.cpp:220) from the
        DEVICE map: every 64^3 chunk the window overlaps is gathered out of the ring buffer by the GPU
        (ws_map_extract_box: chunk layout, x major / z fastest), merged into the global map's chunk and written to
        its file.  The reference downloads the whole window and copies voxel by voxel on the host.
        box_lo / box_hi (inclusive world voxels) restrict the export to a part of the window."""
        self.wait_shift()
        lm, avg = self.local_map_, self.tsdf_.avg_map()
        cs = GlobalMap.CHUNK_SIZE
        with self.mutex_:
            half = lm.size.astype(np.int64) // 2
            lo, hi = lm.pos.astype(np.int64) - half, lm.pos.astype(np.int64) + half
            if box_lo is not None:
                lo = np.maximum(lo, np.asarray(box_lo, dtype=np.int64))
            if box_hi is not None:
                hi = np.minimum(hi, np.asarray(box_hi, dtype=np.int64))
            # one gather per 64-voxel-thick x slab of chunks (a few large device->host copies instead of one small one
            # per chunk); save_box cuts the slab into its chunks
            for cx in range(int(np.floor_divide(lo[0], cs)), int(np.floor_divide(hi[0], cs)) + 1):
                a, b = lo.copy(), hi.copy()
                a[0], b[0] = max(lo[0], cx * cs), min(hi[0], cx * cs + cs - 1)
                lm.map_.save_box(a, b, avg.extract_box(a, b))
            lm.map_.write_back()


class TSDFRegistration(TSDFMapping):
    """cuda::TSDFRegistration (tsdf_registration.cpp:22-96)."""

    def __init__(self, params: Params, local_map: LocalMap, ctx: Context | None = None, flags: int = WS_REG_ALL_POINTS):
        super().__init__(params, local_map, ctx)
        self.reg_ = RegistrationCuda(self.cuda_map_, ctx, flags)
        self.last_iterations = 0

    def register_cloud(self, cloud, pretransform):
        """Returns total_transform (4x4 float32); Gauss-Newton loop on the device."""
        self.reg_.prepare_registration(cloud)
        r, m = self.params_.registration, self.params_.map
        with self.mutex_:
            T, it = self.reg_.register_cloud(self.tsdf_.device_map(), pretransform, r.max_iterations, r.it_weight_gradient,
                                             r.epsilon, m.resolution)
        self.last_iterations = it
        return T
