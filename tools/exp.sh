cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_e -o e -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-registration > /dev/null 2>&1
python tools/rocpd_stats.py gpurun_out/prof_e/e_results.db | grep "march_kernel\|resolve"
rm -rf gpurun_out/prof_e
