cd $GRAFT_REPO_ROOT
python -m warpsense_amd.build --force > /dev/null 2>&1
python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_map_shift.py -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-registration 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
