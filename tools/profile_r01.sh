# rocprofv3 passes for profiles/ (run on the GPU box via gpurun); outputs under gpurun_out/
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_sparse -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}_sparse.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_dense -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --integrate dense > gpurun_out/prof_${TAG}_dense.log 2>&1
# PMC passes in their own runs (kernel-trace only), one counter set per run
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${TAG}_pmc_fetch -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration --integrate dense > gpurun_out/prof_${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${TAG}_pmc_write -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration --integrate dense > gpurun_out/prof_${TAG}_pmc_write.log 2>&1
grep -h '^{' gpurun_out/prof_${TAG}_sparse.log gpurun_out/prof_${TAG}_dense.log | cut -c1-300
ls -la gpurun_out/prof_${TAG}_*/
