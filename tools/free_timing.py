"""Per-workgroup timing of the free pass (a library built with -DWS_FREE_TIMING writes 10 ns ticks and the start tick of every
workgroup into the tail march's statistics slots): how the launch fills the chip and how it ends.
    python -m warpsense_amd.build --variant ftiming "-DWS_FREE_TIMING"
    WS_HIP_LIB=$PWD/warpsense_amd/variants/ftiming.so python tools/free_timing.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warpsense_amd as W  # noqa: E402
from warpsense_amd import synthetic as S  # noqa: E402

tau, res, mw = 1000, 50, 640
lm = W.LocalMap(513, 513, 513, tau, 0)
t = W.TSDFCuda(lm.device_map(), tau, mw, res)
pts = torch.from_numpy(S.os1_128_scan()).cuda()
for _ in range(3):
    t.update_tsdf(pts, (0, 0, 0), (0, 0, 32768))
t.ctx.sync()
words = 3 * 65536 + 8192
out = np.zeros(words, dtype=np.uint32)
rc = t._L.ws_debug_block_stats(t.handle, out.ctypes.data_as(C.c_void_p), words)
assert rc == 0
n = int(os.environ.get("WS_FREE_BLOCKS", "2048"))
dur, start = out[:n].astype(np.int64), out[65536:65536 + n].astype(np.int64)
start = (start - start.min()) & 0xffffffff
end = start + dur
span = end.max()
print(f"workgroups {n}: duration mean {dur.mean() / 100:.1f} us (p10 {np.percentile(dur, 10) / 100:.1f}, p50 {np.percentile(dur, 50) / 100:.1f}, p90 {np.percentile(dur, 90) / 100:.1f}, max {dur.max() / 100:.1f}); "
      f"first start -> last end {span / 100:.1f} us; {dur.sum() / span:.0f} workgroups busy on average; last start at {start.max() / 100:.1f} us")
for frac in (0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
    tt = frac * span
    print(f"  at {100 * frac:.0f} % of the span: {int(((start <= tt) & (end > tt)).sum())} workgroups busy")
late = start > 100  # started after the first wave of workgroups
print(f"  workgroups that started later than 1 us: {int(late.sum())}, their duration mean {dur[late].mean() / 100 if late.any() else 0:.1f} us, the others' {dur[~late].mean() / 100:.1f} us")
