"""Export of the device map to the reference's global-map file, timed on one GPU.

    python tools/bench_export.py [--map 512] [--out /tmp/ws_export.h5]

Prints one JSON line: seconds and GB/s of (a) the whole-window download the reference does before saving
(avg_map().to_host), (b) the chunk gathers alone (ws_map_extract_box per 64^3 chunk), (c) TSDFMapping.write_back
(gathers + merge into chunks + HDF5 writes)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", type=int, default=512)
    ap.add_argument("--out", default="/tmp/ws_export.h5")
    args = ap.parse_args()
    import torch
    import warpsense_amd as W
    from warpsense_amd import synthetic as S
    tau, res = 1000, 50
    size = (args.map,) * 3
    mp = W.MapParams(resolution=res, max_distance=1.0, max_weight=10, size=tuple(s * res / 1000.0 for s in size))
    g = W.GlobalMap(tau, 0, filename=args.out, map_params=mp)
    lm = W.LocalMap(*size, tau, 0, g)
    host_view = lm.device_map()
    tm = W.TSDFMapping(W.Params(mp), lm)
    pts = torch.from_numpy(S.os1_128_scan()).cuda()
    tm.update_tsdf(pts, pos_rm=(0, 0, 0), up_rm=(0, 0, 32768))
    W.pause()
    nbytes = lm.data.nbytes
    t0 = time.perf_counter()
    tm.tsdf().avg_map().to_host(host_view)
    t1 = time.perf_counter()
    avg = tm.tsdf().avg_map()
    half = lm.size.astype(np.int64) // 2
    lo, hi = lm.pos - half, lm.pos + half
    cs = 64
    c0, c1 = np.floor_divide(lo, cs), np.floor_divide(hi, cs)
    n_chunks = 0
    t2 = time.perf_counter()
    for cx in range(c0[0], c1[0] + 1):
        for cy in range(c0[1], c1[1] + 1):
            for cz in range(c0[2], c1[2] + 1):
                base = np.array([cx, cy, cz], dtype=np.int64) * cs
                avg.extract_box(np.maximum(lo, base), np.minimum(hi, base + cs - 1))
                n_chunks += 1
    t3 = time.perf_counter()
    tm.write_back()
    t4 = time.perf_counter()
    g.close()
    print(json.dumps({"map": list(int(s) for s in lm.size), "bytes": nbytes, "chunks": n_chunks,
                      "download_whole_window_s": t1 - t0, "download_GBps": nbytes / (t1 - t0) / 1e9,
                      "chunk_gathers_s": t3 - t2, "chunk_gathers_GBps": nbytes / (t3 - t2) / 1e9,
                      "write_back_to_h5_s": t4 - t3, "write_back_GBps": nbytes / (t4 - t3) / 1e9,
                      "file_bytes": os.path.getsize(args.out)}))


if __name__ == "__main__":
    main()
