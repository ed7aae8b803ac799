#!/bin/bash
# FETCH_SIZE per byte loaded for the load shapes of the scatter's kernels (tools/read_calib.hip)  ->  gpurun_out/TAG_read_calib.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
mkdir -p gpurun_out
[ -x tools/read_calib.out ] || hipcc --offload-arch=gfx950 -O3 tools/read_calib.hip -o tools/read_calib.out
rm -rf gpurun_out/prof_${TAG}_rcal
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${TAG}_rcal -o pmc -- tools/read_calib.out > gpurun_out/prof_${TAG}_rcal.log 2>&1
DB=$(ls gpurun_out/prof_${TAG}_rcal/*.db gpurun_out/prof_${TAG}_rcal/*/*.db 2>/dev/null | head -1)
python - $DB > gpurun_out/${TAG}_read_calib.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, avg(v), avg(d) from (select name, dispatch_id, sum(counter_value) as v, avg(duration) as d from pmc_events "
                  "where counter_name = 'FETCH_SIZE' group by name, dispatch_id) group by name").fetchall()
N = 1 << 22
loaded = {"rcal_stream16": 16 * N, "rcal_stream8": 8 * N, "rcal_stream4": 4 * N, "rcal_run16_256": 16 * N, "rcal_run8_256": 8 * N, "rcal_scatter8": 8 * N, "rcal_scatter1": N}
print(f"{'kernel':<18} {'loads':>9} {'bytes loaded':>13} {'FETCH_SIZE bytes':>17} {'counted/loaded':>15} {'counted per load':>17} {'us':>8}")
for name, kb, dur in sorted(rows):
    key = name.split("(")[0]
    if key in loaded:
        b = kb * 1024.0
        print(f"{key:<18} {N:>9} {loaded[key]:>13} {b:>17.0f} {b / loaded[key]:>15.2f} {b / N:>17.1f} {dur / 1e3:>8.1f}")
PY
cat gpurun_out/${TAG}_read_calib.txt
rm -rf gpurun_out/prof_${TAG}_rcal
