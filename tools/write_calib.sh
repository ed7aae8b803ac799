#!/bin/bash
# WRITE_SIZE per byte stored for the store shapes of the tail march (tools/write_calib.hip)  ->  gpurun_out/TAG_write_calib.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
mkdir -p gpurun_out
[ -x tools/write_calib.out ] || hipcc --offload-arch=gfx950 -O3 tools/write_calib.hip -o tools/write_calib.out
rm -rf gpurun_out/prof_${TAG}_wcal
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${TAG}_wcal -o pmc -- tools/write_calib.out > gpurun_out/prof_${TAG}_wcal.log 2>&1
DB=$(ls gpurun_out/prof_${TAG}_wcal/*.db gpurun_out/prof_${TAG}_wcal/*/*.db 2>/dev/null | head -1)
python - $DB > gpurun_out/${TAG}_write_calib.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, avg(v), avg(d) from (select name, dispatch_id, sum(counter_value) as v, avg(duration) as d from pmc_events "
                  "where counter_name = 'WRITE_SIZE' group by name, dispatch_id) group by name").fetchall()
N = 1 << 22
stored = {"calib_stream16": 16 * N, "calib_rec8_dense": 8 * N, "calib_rec8_scatter": 8 * N, "calib_rec8_revisit": 8 * N, "calib_byte_scatter": N, "calib_byte_dense": N}
print(f"{'kernel':<22} {'stores':>9} {'bytes stored':>13} {'WRITE_SIZE bytes':>17} {'counted/stored':>15} {'counted per store':>18} {'us':>8}")
for name, kb, dur in sorted(rows):
    key = name.split("(")[0]
    if key in stored:
        b = kb * 1024.0
        print(f"{key:<22} {N:>9} {stored[key]:>13} {b:>17.0f} {b / stored[key]:>15.2f} {b / N:>18.1f} {dur / 1e3:>8.1f}")
PY
cat gpurun_out/${TAG}_write_calib.txt
