cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_tsdf.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-registration 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(sum(v['avg_us'] for v in d['kernels'].values()),1))"; done
