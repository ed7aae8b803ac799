# rocprofv3 passes for profiles/ (run on the GPU box via gpurun); outputs under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_sparse -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}_sparse.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_dense -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --integrate dense > gpurun_out/prof_${TAG}_dense.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_tiles -o trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --scatter tiles > gpurun_out/prof_${TAG}_tiles.log 2>&1
# PMC passes in their own runs (kernel-trace only), one counter per run
for mode in sparse dense; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${TAG}_pmc_fetch_${mode} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration --integrate ${mode} > gpurun_out/prof_${TAG}_pmc_fetch_${mode}.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${TAG}_pmc_write_${mode} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration --integrate ${mode} > gpurun_out/prof_${TAG}_pmc_write_${mode}.log 2>&1
done
cat gpurun_out/bench_${TAG}.json | cut -c1-400
