"""What a workgroup's busy time in the tile resolve is made of (a library built with -DWS_RESOLVE_TIMING=2 writes its busy time
and sub-chunks << 16 | contested voxels): least-squares fit  busy = a * tiles + b * sub-chunks + c * contested voxels.
   WS_HIP_LIB=warpsense_amd/variants/NAME.so python tools/resolve_fit.py"""
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
subs = (r[:n, 1] >> 16).astype(np.float64)
cont = (r[:n, 1] & 0xffff).astype(np.float64)
print(f"workgroups {n}: busy mean {busy.mean():.1f} us (min {busy.min():.1f}, max {busy.max():.1f}, sd {busy.std():.1f}); sub-chunks mean {subs.mean():.0f} (sd {subs.std():.0f}), "
      f"contested voxels mean {cont.mean():.0f} (sd {cont.std():.0f})")
for name, cols in (("sub-chunks", [subs]), ("contested", [cont]), ("sub-chunks + contested", [subs, cont])):
    A = np.stack([np.ones(n)] + cols, axis=1)
    coef, res_, *_ = np.linalg.lstsq(A, busy, rcond=None)
    pred = A @ coef
    print(f"  busy ~ const + {name}: coefficients {np.round(coef, 4)}, residual sd {np.std(busy - pred):.2f} us (of {busy.std():.2f})")
print("  correlation busy / sub-chunks %.3f, busy / contested %.3f, sub-chunks / contested %.3f" % (np.corrcoef(busy, subs)[0, 1], np.corrcoef(busy, cont)[0, 1], np.corrcoef(subs, cont)[0, 1]))
