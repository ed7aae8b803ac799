# rocprofv3 passes for profiles/ (run on the GPU box via gpurun); outputs under gpurun_out/
#   bash tools/profile_r02.sh TAG [quick]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
QUICK=${2:-}
for mode in sparse dense; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_${mode} -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --integrate ${mode} > gpurun_out/prof_${TAG}_${mode}.log 2>&1
  python tools/rocpd_stats.py gpurun_out/prof_${TAG}_${mode}/*/trace_results.db > gpurun_out/${TAG}_kernel_stats_${mode}.txt 2>/dev/null || python tools/rocpd_stats.py $(ls gpurun_out/prof_${TAG}_${mode}/*.db gpurun_out/prof_${TAG}_${mode}/*/*.db 2>/dev/null | head -1) > gpurun_out/${TAG}_kernel_stats_${mode}.txt
done
if [ -z "$QUICK" ]; then
  python bench.py --steps 20 --warmup 3 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
  # PMC passes in their own runs (kernel-trace only), one counter per run
  for mode in sparse dense; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc ${ctr} -d gpurun_out/prof_${TAG}_pmc_${ctr}_${mode} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration --integrate ${mode} > gpurun_out/prof_${TAG}_pmc_${ctr}_${mode}.log 2>&1
      python tools/pmc_summary.py $(ls gpurun_out/prof_${TAG}_pmc_${ctr}_${mode}/*.db gpurun_out/prof_${TAG}_pmc_${ctr}_${mode}/*/*.db 2>/dev/null | head -1) > gpurun_out/${TAG}_pmc_${ctr}_${mode}.txt
    done
  done
fi
head -20 gpurun_out/${TAG}_kernel_stats_sparse.txt
