for e in 4 8 16; do
  WS_EXTRA_FLAGS=-DWS_COLLECT_LANES=$e python -m warpsense_amd.build --force > /dev/null 2>&1
  echo "COLLECT_LANES $e"; python -m pytest tests/test_gpu_tsdf.py -q 2>&1 | tail -1
  python bench.py --no-cpu-baseline --no-registration --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done
