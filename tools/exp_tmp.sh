#!/bin/bash
WS_EXTRA_FLAGS="-DWS_REG_TIMING" python -m warpsense_amd.build --force 2>&1 | grep -i " error" 
python bench.py --steps 2 --warmup 1 2>&1 | grep "reg_loop wg" | tail -2
python -m warpsense_amd.build --force > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_gpu_registration.py tests/test_gpu_replay.py -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
