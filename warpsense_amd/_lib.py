"""ctypes binding of libwarpsense_hip.so — the C ABI declared in include/warpsense_hip.h.

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# WS_HIP_LIB: another build of the same library (A/B measurements of kernel variants on one box; tools/ab_bench.sh)
LIB_PATH = os.environ.get("WS_HIP_LIB") or os.path.join(PKG_DIR, "libwarpsense_hip.so")

WS_MAP_AVG, WS_MAP_NEW = 0, 1
WS_INTEGRATE_SPARSE, WS_INTEGRATE_DENSE, WS_INTEGRATE_SPARSE_SEPARATE = 0, 1, 2
WS_REG_ALL_POINTS, WS_REG_COMPAT_REFERENCE_LAUNCH = 0, 1
WS_REG_LOOP_RESIDENT, WS_REG_LOOP_LAUNCHES = 0, 1
(WS_K_SETUP, WS_K_MARCH_TAILS, WS_K_MARCH_FREE, WS_K_TILE_BIN, WS_K_TILE_RESOLVE, WS_K_INTEGRATE, WS_K_REG, WS_K_UPDATE) = range(8)
KERNEL_CLASSES = ["ray_setup", "march_tails", "march_free", "tile_bin", "tile_resolve", "integrate", "reg_iteration", "tsdf_update"]

# every symbol include/warpsense_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "ws_last_error", "ws_version", "ws_ctx_create", "ws_ctx_destroy", "ws_ctx_set_stream", "ws_sync",
    "ws_device_reset", "ws_map_create", "ws_map_destroy", "ws_map_upload", "ws_map_set_params", "ws_map_download",
    "ws_map_extract_box", "ws_map_insert_box", "ws_shift_begin", "ws_shift_reserve", "ws_shift_count", "ws_shift_entering", "ws_shift_wait", "ws_shift_slab",
    "ws_shift_end", "ws_map_get_params", "ws_map_device_data", "ws_map_n_voxels", "ws_tsdf_update", "ws_tsdf_update_dev", "ws_tsdf_scatter_dev",
    "ws_tsdf_integrate", "ws_tsdf_set_integrate", "ws_tsdf_set_capacity", "ws_debug_tsdf_chunk_policy", "ws_tsdf_stats", "ws_reg_create", "ws_reg_destroy", "ws_reg_prepare",
    "ws_reg_prepare_dev", "ws_reg_points_dev", "ws_reg_iterate", "ws_register_cloud", "ws_reg_begin", "ws_reg_accumulate_dev",
    "ws_reg_solve_dev", "ws_reg_iterate_shard_dev", "ws_reg_poll", "ws_reg_peer_mailbox", "ws_reg_peer_connect", "ws_reg_peer_connect_local",
    "ws_reg_peer_disconnect", "ws_reg_peer_reset", "ws_register_cloud_peers", "ws_reg_set_loop", "ws_debug_solve6", "ws_debug_reg_stall", "ws_debug_reg_server", "ws_debug_reg_mail_selftest", "ws_debug_reg_sums", "ws_debug_block_stats", "ws_scan_create", "ws_scan_destroy", "ws_scan_preprocess",
    "ws_scan_preprocess_dev", "ws_scan_points_dev", "ws_scan_download", "ws_prof_enable", "ws_prof_read", "ws_prof_reset",
]


class WsError(RuntimeError):
    pass


H5_LIB_PATH = os.path.join(PKG_DIR, "libwarpsense_h5.so")
H5_EXPORTS = ["ws_h5_last_error", "ws_h5_create", "ws_h5_open", "ws_h5_close", "ws_h5_flush", "ws_h5_write_meta", "ws_h5_read_meta",
              "ws_h5_write_chunk", "ws_h5_read_chunk", "ws_h5_num_chunks", "ws_h5_list_chunks", "ws_h5_write_pose", "ws_h5_num_poses",
              "ws_h5_read_pose"]
_h5 = None


def h5_available() -> bool:
    return os.path.exists(H5_LIB_PATH)


def load_h5() -> C.CDLL:
    """Load libwarpsense_h5.so (include/warpsense_h5.h); raises if it was not built (no HDF5 C library on the box)."""
    global _h5
    if _h5 is not None:
        return _h5
    if not os.path.exists(H5_LIB_PATH):
        raise WsError(f"{H5_LIB_PATH} is missing: `python -m warpsense_amd.build` builds it where the HDF5 C library is installed")
    L = C.CDLL(H5_LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    P = C.POINTER
    L.ws_h5_last_error.restype = C.c_char_p
    L.ws_h5_create.argtypes = [C.c_char_p, P(vp)]
    L.ws_h5_open.argtypes = [C.c_char_p, C.c_int, P(vp)]
    L.ws_h5_close.argtypes = [vp]
    L.ws_h5_flush.argtypes = [vp]
    L.ws_h5_write_meta.argtypes = [vp, i32, vp, C.c_float, i32, i32]
    L.ws_h5_read_meta.argtypes = [vp, P(i32), vp, P(C.c_float), P(i32), P(i32)]
    L.ws_h5_write_chunk.argtypes = [vp, i32, i32, i32, vp]
    L.ws_h5_read_chunk.argtypes = [vp, i32, i32, i32, vp, P(i32)]
    L.ws_h5_num_chunks.argtypes = [vp, P(i64)]
    L.ws_h5_list_chunks.argtypes = [vp, vp, i64, P(i64)]
    L.ws_h5_write_pose.argtypes = [vp, vp]
    L.ws_h5_num_poses.argtypes = [vp, P(i64)]
    L.ws_h5_read_pose.argtypes = [vp, i64, vp]
    _h5 = L
    return L


def check_h5(rc: int, what: str):
    if rc != 0:
        raise WsError(f"{what}: {load_h5().ws_h5_last_error().decode(errors='replace')}")


class TsdfStats(C.Structure):
    _fields_ = [("contested_voxels", C.c_int64), ("records", C.c_int64), ("tiles", C.c_int64),
                ("error_flags", C.c_int32), ("hash_entries", C.c_int32), ("runs", C.c_int64), ("free_space_hits", C.c_int64),
                ("record_slots", C.c_int64), ("record_capacity", C.c_int64)]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raises if it has not been built (python -m warpsense_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WsError(f"{LIB_PATH} is missing: build it with `python -m warpsense_amd.build` "
                      "(there is no CPU fallback for the hot path)")
    # PyTorch-ROCm ships its own libamdhip64; two HIP runtimes in one process do not share the device (whichever
    # initialises second reports "no ROCm-capable device").  Load torch's first, so that this library binds to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, sz, i64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t, C.c_int64
    P = C.POINTER
    L.ws_last_error.restype = C.c_char_p
    L.ws_version.restype = C.c_int
    L.ws_ctx_create.argtypes = [C.c_int, P(vp)]
    L.ws_ctx_destroy.argtypes = [vp]
    L.ws_ctx_set_stream.argtypes = [vp, vp]
    L.ws_sync.argtypes = [vp]
    L.ws_map_create.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, P(vp)]
    L.ws_map_destroy.argtypes = [vp]
    L.ws_map_upload.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.ws_map_set_params.argtypes = [vp, C.c_int, vp, vp, vp]
    L.ws_map_download.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.ws_map_extract_box.argtypes = [vp, C.c_int, vp, vp, vp]
    L.ws_map_insert_box.argtypes = [vp, C.c_int, vp, vp, vp]
    L.ws_map_get_params.argtypes = [vp, C.c_int, vp, vp, vp]
    L.ws_shift_begin.argtypes = [vp, vp, u32, P(vp)]
    L.ws_shift_count.argtypes = [vp]
    L.ws_shift_reserve.argtypes = [vp, C.c_uint64]
    L.ws_shift_entering.argtypes = [vp, C.c_int, vp, vp]
    L.ws_shift_wait.argtypes = [vp]
    L.ws_shift_slab.argtypes = [vp, C.c_int, vp, vp, P(vp)]
    L.ws_shift_end.argtypes = [vp]
    L.ws_map_device_data.argtypes = [vp, C.c_int]
    L.ws_map_device_data.restype = vp
    L.ws_map_n_voxels.argtypes = [vp]
    L.ws_map_n_voxels.restype = i64
    L.ws_tsdf_update.argtypes = [vp, vp, sz, vp, vp]
    L.ws_tsdf_update_dev.argtypes = [vp, vp, sz, vp, vp]
    L.ws_tsdf_scatter_dev.argtypes = [vp, vp, sz, vp, vp]
    L.ws_tsdf_integrate.argtypes = [vp]
    L.ws_tsdf_set_integrate.argtypes = [vp, C.c_int]
    L.ws_tsdf_set_capacity.argtypes = [vp, C.c_uint64]
    L.ws_tsdf_stats.argtypes = [vp, P(TsdfStats)]
    L.ws_reg_create.argtypes = [vp, sz, P(vp)]
    L.ws_reg_destroy.argtypes = [vp]
    L.ws_reg_prepare.argtypes = [vp, vp, sz]
    L.ws_reg_prepare_dev.argtypes = [vp, vp, sz]
    L.ws_reg_points_dev.argtypes = [vp, P(sz)]
    L.ws_reg_points_dev.restype = vp
    L.ws_reg_iterate.argtypes = [vp, vp, vp, i32, u32, vp, vp, P(i32), P(i32)]
    L.ws_register_cloud.argtypes = [vp, vp, vp, i32, C.c_float, C.c_float, i32, u32, vp, P(i32)]
    L.ws_reg_begin.argtypes = [vp, vp, i32, C.c_float, C.c_float]
    L.ws_reg_set_loop.argtypes = [vp, C.c_int]
    L.ws_reg_iterate_shard_dev.argtypes = [vp, vp, i32, u32, sz, sz, vp, i32]
    L.ws_debug_solve6.argtypes = [vp, vp, vp, sz, vp, vp]
    L.ws_debug_reg_stall.argtypes = [vp, C.c_int32, vp]
    L.ws_debug_reg_server.argtypes = [vp, C.c_int32, C.c_int32, vp]
    L.ws_debug_reg_mail_selftest.argtypes = []
    L.ws_debug_reg_sums.argtypes = [vp, vp]
    L.ws_debug_block_stats.argtypes = [vp, vp, sz]
    L.ws_debug_tsdf_chunk_policy.argtypes = [vp, C.c_uint64, u32]
    L.ws_scan_create.argtypes = [vp, sz, P(vp)]
    L.ws_scan_destroy.argtypes = [vp]
    L.ws_scan_preprocess.argtypes = [vp, vp, sz, sz, vp, i32, P(sz)]
    L.ws_scan_preprocess_dev.argtypes = [vp, vp, sz, sz, vp, i32, P(sz)]
    L.ws_scan_points_dev.argtypes = [vp]
    L.ws_scan_points_dev.restype = vp
    L.ws_scan_download.argtypes = [vp, vp, sz, P(sz)]
    L.ws_reg_accumulate_dev.argtypes = [vp, vp, i32, u32, sz, sz, vp]
    L.ws_reg_solve_dev.argtypes = [vp, vp]
    L.ws_reg_poll.argtypes = [vp, P(i32), P(i32), vp]
    L.ws_reg_peer_mailbox.argtypes = [vp, vp]
    L.ws_reg_peer_connect.argtypes = [vp, i32, i32, vp, i32]
    L.ws_reg_peer_connect_local.argtypes = [vp, i32, i32, vp, i32]
    L.ws_reg_peer_disconnect.argtypes = [vp]
    L.ws_reg_peer_reset.argtypes = [vp]
    L.ws_register_cloud_peers.argtypes = [vp, vp, sz, sz, vp, i32, C.c_float, C.c_float, i32, u32, vp, P(i32)]
    L.ws_prof_enable.argtypes = [vp, u32]
    L.ws_prof_read.argtypes = [vp, C.c_int, P(C.c_double), P(i64)]
    L.ws_prof_reset.argtypes = [vp]
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ws_last_error()
        raise WsError(f"{what} failed with status {rc}: {msg.decode(errors='replace') if msg else ''}")
