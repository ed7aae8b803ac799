#!/bin/bash
# Lane utilisation of the update kernels (VERDICT r4 #1a): SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) = the share of the
# 64 lanes of an issued vector instruction that the EXEC mask leaves on.  One --pmc pass with --kernel-trace only.
#   bash tools/pmc_lanes.sh TAG      -> gpurun_out/TAG_pmc_lanes.txt
# NOTE: lanes that are on but compute something that is thrown away (the branch-free sample phase of the marches keeps lanes
# stepping whose samples are used up) count as active here; tools/lane_model.py counts the USEFUL lanes from the scan itself.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WS_BENCH_SKIP_SHARDED=1 WS_BENCH_SKIP_DENSE_EQ=1
TAG=${1:-r05}
mkdir -p gpurun_out
rm -rf gpurun_out/prof_${TAG}_pmc_lanes
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
  -d gpurun_out/prof_${TAG}_pmc_lanes -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration > gpurun_out/prof_${TAG}_pmc_lanes.log 2>&1
DB=$(ls gpurun_out/prof_${TAG}_pmc_lanes/*.db gpurun_out/prof_${TAG}_pmc_lanes/*/*.db 2>/dev/null | head -1)
python tools/pmc_summary.py $DB | grep -E "^kernel|march|resolve|ray_s" > gpurun_out/${TAG}_pmc_lanes.txt
python - $DB >> gpurun_out/${TAG}_pmc_lanes.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, avg(v) from (select name, counter_name, dispatch_id, sum(counter_value) as v from pmc_events "
                  "group by name, counter_name, dispatch_id) group by name, counter_name").fetchall()
by = {}
for name, c, v in rows:
    by.setdefault(name, {})[c] = v
print("\nlanes active per issued vector instruction (EXEC mask): SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)")
for name, d in sorted(by.items()):
    if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        if any(k in name for k in ("march", "resolve", "ray_s")):
            print(f"{name[:70]:<70} {100.0 * d['SQ_THREAD_CYCLES_VALU'] / (64.0 * d['SQ_ACTIVE_INST_VALU']):6.1f} %   "
                  f"({d['SQ_INSTS_VALU'] / 1e6:.1f} M vector instructions per launch)")
PY
cat gpurun_out/${TAG}_pmc_lanes.txt
