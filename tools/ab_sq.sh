#!/bin/bash
# SQ counters per library variant on ONE box (warpsense_amd/variants/*.so, see tools/ab_bench.sh): vector / scalar instructions,
# busy and waiting wave cycles of the update's kernels.    bash tools/ab_sq.sh
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
export WS_BENCH_SKIP_SHARDED=1 WS_BENCH_SKIP_DENSE_EQ=1 WS_BENCH_SKIP_DROPIN=1
mkdir -p gpurun_out
for so in warpsense_amd/variants/*.so; do
  name=$(basename $so .so)
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    rm -rf gpurun_out/prof_sq_$name
    WS_HIP_LIB=$PWD/$so rocprofv3 --kernel-trace --pmc ${grp} -d gpurun_out/prof_sq_$name -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration > gpurun_out/prof_sq_$name.log 2>&1
    echo "== $name"
    python tools/pmc_summary.py $(ls gpurun_out/prof_sq_$name/*.db gpurun_out/prof_sq_$name/*/*.db 2>/dev/null | head -1) | grep -E "^kernel|march|resolve_kernel<false, true|ray_s" | cut -c1-260 | tee gpurun_out/sq_$name.txt
    rm -rf gpurun_out/prof_sq_$name
  done
done
