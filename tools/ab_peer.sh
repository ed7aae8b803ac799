cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do for so in warpsense_amd/variants/*.so; do name=$(basename "$so" .so); echo "$name: $(WS_HIP_LIB="$PWD/$so" python tools/peer_bench.py --ranks 2 --reps 20 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["us_per_iteration"], d["one_rank_us_per_iteration"], d["same_result_as_one_rank"])')"; done; done
