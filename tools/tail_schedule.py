#!/usr/bin/env python3
"""How well the tail march fills the chip: per-workgroup start / end times (library built with -DWS_TAIL_TIMING, selected with
WS_HIP_LIB) of one update of the benchmark scan -> makespan against the sum of the workgroups' durations over the slots the chip
has, and where the long workgroups sit in the launch order."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    tau, mw, res = 1000, 640, 50
    ctx = W.Context(0)
    lm = W.LocalMap(512, 512, 512, tau, 0)
    host_map = lm.device_map()
    host_map.data_ = None
    tsdf = W.TSDFCuda(host_map, tau, mw, res, ctx)
    d = torch.from_numpy(S.os1_128_scan()).cuda()
    for _ in range(3):
        tsdf.update_tsdf(d, (0, 0, 0), (0, 0, 32768))
    ctx.sync()
    words = 65536
    out = np.zeros(words, dtype=np.uint32)
    rc = tsdf._L.ws_debug_block_stats(tsdf.handle, out.ctypes.data_as(C.c_void_p), words)
    assert rc == 0
    n = 4096
    rec = out[:n].astype(np.int64)
    t0, tm, t1 = (out[k:k + n].astype(np.int64) for k in (16384, 32768, 49152))
    ok = t1 > 0
    base = t0[ok].min()
    t0, tm, t1 = (t0 - base) / 100.0, (tm - base) / 100.0, (t1 - base) / 100.0  # us
    dur = t1 - t0
    p1, p2 = tm - t0, t1 - tm
    slots = 6 * 256
    res_ = {"workgroups": int(ok.sum()), "makespan_us": float(t1[ok].max()), "sum_dur_us": float(dur[ok].sum()), "ideal_us": float(dur[ok].sum() / slots),
            "mean_dur_us": float(dur[ok].mean()), "max_dur_us": float(dur[ok].max()), "p95_dur_us": float(np.percentile(dur[ok], 95)),
            "phase1_mean_us": float(p1[ok].mean()), "phase2_mean_us": float(p2[ok].mean()),
            "records_mean": float(rec.mean()), "records_max": int(rec.max()),
            "corr_dur_records": float(np.corrcoef(dur[ok], rec[ok])[0, 1]),
            "last_start_us": float(t0[ok].max()),
            "dur_of_last_10pct_started_us": float(dur[ok][np.argsort(t0[ok])[-n // 10:]].mean())}
    # active workgroups over time
    edges = np.linspace(0, res_["makespan_us"], 21)
    act = [int(((t0[ok] <= e) & (t1[ok] > e)).sum()) for e in edges]
    res_["active_at_20ths"] = act
    print(json.dumps(res_))


if __name__ == "__main__":
    main()
