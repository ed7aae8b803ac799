#!/bin/bash
# SQ counter passes of the update kernels (two groups of eight, --kernel-trace only): where the cycles of the march / resolve go.
#   bash tools/pmc_sq.sh TAG      -> gpurun_out/TAG_pmc_sq{1,2}.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WS_BENCH_SKIP_SHARDED=1
TAG=${1:-r04}
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rm -rf gpurun_out/prof_${TAG}_pmc_sq${i}
  rocprofv3 --kernel-trace --pmc ${grp} -d gpurun_out/prof_${TAG}_pmc_sq${i} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration > gpurun_out/prof_${TAG}_pmc_sq${i}.log 2>&1
  python tools/pmc_summary.py $(ls gpurun_out/prof_${TAG}_pmc_sq${i}/*.db gpurun_out/prof_${TAG}_pmc_sq${i}/*/*.db 2>/dev/null | head -1) | grep -E "^kernel|march|resolve|ray_s" > gpurun_out/${TAG}_pmc_sq${i}.txt
done
cat gpurun_out/${TAG}_pmc_sq1.txt gpurun_out/${TAG}_pmc_sq2.txt
