"""Per-workgroup timing of the tail march (a library built with -DWS_TAIL_TIMING writes 10 ns ticks of the march and of the
flush of every work item into the statistics slots, and the wall-clock tick it started at).

    python -m warpsense_amd.build --variant timing "-DWS_TAIL_TIMING"
    WS_HIP_LIB=$PWD/warpsense_amd/variants/timing.so python tools/tail_timing.py
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
n = 4096
march, flush, start = out[:n].astype(np.int64), out[65536:65536 + n].astype(np.int64), out[2 * 65536 + 8192:2 * 65536 + 8192 + n].astype(np.int64)
start = (start - start.min()) & 0xffffffff
end = start + march + flush
print(f"items {n}: march mean {march.mean() / 100:.1f} us (max {march.max() / 100:.1f}), flush mean {flush.mean() / 100:.1f} us (max {flush.max() / 100:.1f}); "
      f"first start -> last end {end.max() / 100:.1f} us; sum of durations / span = {(march + flush).sum() / end.max():.0f} workgroups busy on average")
# how the launch ends: workgroups still busy at the given fraction of the span
dur = march + flush
for frac in (0.5, 0.75, 0.85, 0.9, 0.95):
    tt = frac * end.max()
    print(f"  at {100 * frac:.0f} % of the span: {int(((start <= tt) & (end > tt)).sum())} workgroups busy")
print(f"  item duration: p10 {np.percentile(dur, 10) / 100:.1f} us, p50 {np.percentile(dur, 50) / 100:.1f}, p90 {np.percentile(dur, 90) / 100:.1f}, max {dur.max() / 100:.1f}; last item starts at {start.max() / 100:.1f} us")
