"""The two phases of a workgroup of the tile resolve (a library built with -DWS_RESOLVE_TIMING=3): the listed tiles (records) and
the scan of the flag planes with the unlisted tiles it finds (free-space marks only).
   WS_HIP_LIB=warpsense_amd/variants/NAME.so python tools/resolve_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import warpsense_amd as W
from warpsense_amd import synthetic as S
import torch

tau, res = 1000, 50
lm = W.LocalMap(512, 512, 512, tau, 0)
t = W.TSDFCuda(lm.device_map(), tau, 640, res)
pts = torch.from_numpy(S.os1_128_scan()).cuda()
for _ in range(3):
    t.update_tsdf(pts, (0, 0, 0), (0, 0, 32768))
t.ctx.sync()
buf = np.zeros(2 * 65536 + 8192, dtype=np.uint32)
t._L.ws_debug_block_stats(t.handle, buf.ctypes.data_as(C.c_void_p), buf.size)
r = buf[2 * 65536:2 * 65536 + 8192].reshape(-1, 2)
busy = r[:, 0].astype(np.float64) / 100.0
n = int((busy > 0).sum())
busy = busy[:n]
listed = (r[:n, 1] & 0xfffff).astype(np.float64) / 100.0
unl = (r[:n, 1] >> 20).astype(np.float64)
rest = busy - listed
print(f"workgroups {n}: busy {busy.mean():.1f} us (sd {busy.std():.1f}, max {busy.max():.1f}) = listed tiles {listed.mean():.1f} (sd {listed.std():.1f}, min {listed.min():.1f}, max {listed.max():.1f}) "
      f"+ flag scan and unlisted tiles {rest.mean():.1f} (sd {rest.std():.1f}, min {rest.min():.1f}, max {rest.max():.1f}); unlisted tiles per workgroup {unl.mean():.1f} (sd {unl.std():.1f}, max {unl.max():.0f}), {int(unl.sum())} in all")
print("  correlation (flag-scan phase, unlisted tiles found) %.3f; (listed phase, flag-scan phase) %.3f" % (np.corrcoef(rest, unl)[0, 1], np.corrcoef(listed, rest)[0, 1]))
A = np.stack([np.ones(n), unl], axis=1)
coef, *_ = np.linalg.lstsq(A, rest, rcond=None)
print(f"  flag-scan phase ~ {coef[0]:.2f} us + {coef[1]:.3f} us per unlisted tile")
