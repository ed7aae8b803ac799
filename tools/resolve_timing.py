"""Per-workgroup busy time of the tile resolve (a library built with -DWS_RESOLVE_TIMING): how evenly the tiles are dealt out.
   WS_HIP_LIB=warpsense_amd/variants/NAME.so python tools/resolve_timing.py"""
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
busy, start = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
n = int((busy > 0).sum())
busy, start = busy[:n], start[:n]
start = (start - start.min()) & 0xffffffff
end = start + busy
print(f"workgroups {n}: busy mean {busy.mean() / 100:.1f} us, min {busy.min() / 100:.1f}, max {busy.max() / 100:.1f}; "
      f"last start {start.max() / 100:.1f} us, kernel span {end.max() / 100:.1f} us; p50 end {np.percentile(end, 50) / 100:.1f}, p90 {np.percentile(end, 90) / 100:.1f}, p99 {np.percentile(end, 99) / 100:.1f}")
# by XCD (workgroup b runs on XCD b % 8 under round-robin dispatch) and by CU slot
for x in range(8):
    sel = busy[x::8]
    print(f"  XCD {x}: busy mean {sel.mean() / 100:.1f} us, max {sel.max() / 100:.1f}")
