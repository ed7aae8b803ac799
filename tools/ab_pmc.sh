cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WS_BENCH_SKIP_SHARDED=1
bash tools/ab_bench.sh --no-registration 2>&1 | tail -3
for so in warpsense_amd/variants/*.so; do
  name=$(basename $so .so)
  for ctr in WRITE_SIZE; do
    rm -rf gpurun_out/prof_ab_$name
    WS_HIP_LIB=$PWD/$so rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/prof_ab_$name -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration > /dev/null 2>&1
    echo $name $ctr; python tools/pmc_summary.py $(ls gpurun_out/prof_ab_$name/*.db gpurun_out/prof_ab_$name/*/*.db 2>/dev/null | head -1) | grep -E "march|resolve" | cut -c1-150
    rm -rf gpurun_out/prof_ab_$name
  done
done
