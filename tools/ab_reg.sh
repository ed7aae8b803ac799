#!/bin/bash
# A/B of library variants on ONE box for the registration loop: tools/reg_fit.py once per warpsense_amd/variants/*.so
#   python -m warpsense_amd.build --variant NAME "-DFLAG=.."   (here), then on the GPU box:  bash tools/ab_reg.sh [reg_fit args]
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for so in warpsense_amd/variants/*.so; do
  name=$(basename "$so" .so)
  echo "$name: $(WS_HIP_LIB="$PWD/$so" python tools/reg_fit.py "$@" 2>&1 | tail -1)"
done
done
