"""warpsense_amd — MI355X-native TSDF update and Point-to-TSDF registration (the hot path of warpsense).

The compute lives in libwarpsense_hip.so (warpsense_amd/csrc, C ABI in include/warpsense_hip.h);
this package is the thin host-side mirror of the reference's device API plus the synthetic workload.
"""
from .api import (Context, DeviceMap, DeviceMapMemWrapper, DevicePoints, GlobalMap, LocalMap, MapParams, Params, RegistrationCuda,  # noqa: F401
                  RegistrationParams, ScanPreprocessor, TSDFCuda, TSDFMapping, TSDFRegistration, cleanup, pack_entry, pause, pose_to_values, to_int_mat,
                  to_map, unpack_entry)
from ._lib import (WS_INTEGRATE_DENSE, WS_INTEGRATE_SPARSE, WS_INTEGRATE_SPARSE_SEPARATE, WS_MAP_AVG, WS_MAP_NEW, WS_REG_ALL_POINTS,  # noqa: F401
                   WS_REG_COMPAT_REFERENCE_LAUNCH, WS_REG_LOOP_LAUNCHES, WS_REG_LOOP_RESIDENT, WsError)

__all__ = [n for n in dir() if not n.startswith("_")]
from .app import App  # noqa: F401,E402
