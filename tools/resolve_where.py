"""Busy time of the tile resolve's workgroups by WHERE they ran (a library built with -DWS_RESOLVE_TIMING=4 writes HW_ID / XCC_ID):
how much of the spread is between compute units and how much inside one.
   WS_HIP_LIB=warpsense_amd/variants/NAME.so python tools/resolve_where.py"""
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
w = r[:n, 1]
xcc, hw = (w >> 16) & 0xf, w & 0xffff
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
key = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu
ids, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
means = np.bincount(inv, busy) / cnt
within = busy - means[inv]
print(f"workgroups {n} on {len(ids)} compute units ({cnt.min()}-{cnt.max()} each): busy mean {busy.mean():.1f} us, sd {busy.std():.1f}; "
      f"sd of the compute units' means {means.std():.1f}, sd inside a compute unit {within.std():.1f}")
for name, k in (("XCD", xcc), ("shader engine", xcc.astype(np.int64) * 8 + se), ("CU number inside its array", cu)):
    u, iv, c = np.unique(k, return_inverse=True, return_counts=True)
    m = np.bincount(iv, busy) / c
    print(f"  by {name}: " + " ".join(f"{x:.0f}" for x in m))
# which workgroups share a compute unit (dispatch order)
if len(ids) > 1:
    for c0 in (0, len(ids) // 2):
        sel = np.where(key == ids[c0])[0]
        print(f"  compute unit {c0}: workgroups {sel.tolist()}, busy {np.round(busy[sel], 1).tolist()}")
else:
    print("  (not a -DWS_RESOLVE_TIMING=4 build: no placement words)")
