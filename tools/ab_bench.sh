#!/bin/bash
# A/B of library variants on ONE box (the boxes differ by a few % between gpurun calls):
#   python -m warpsense_amd.build --variant NAME "-DFLAG=.."   (here), then on the GPU box:  bash tools/ab_bench.sh [bench args]
# runs bench.py once per warpsense_amd/variants/*.so (WS_HIP_LIB) and prints update span / per-kernel-class times.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for so in warpsense_amd/variants/*.so; do
  name=$(basename "$so" .so)
  WS_HIP_LIB="$PWD/$so" python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" > gpurun_out/ab_${name}.json 2> gpurun_out/ab_${name}.err || { echo "$name FAILED"; tail -n 5 gpurun_out/ab_${name}.err; continue; }
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
d = json.load(open(f"gpurun_out/ab_{name}.json"))
k = {a: round(b["avg_us"], 1) for a, b in d["kernels"].items()}
print(f"{name:28s} scans/s {d['value']:7.1f}  update span {d['roofline']['avg_launch_us']:6.1f} us  classes {k}")
PY
done
