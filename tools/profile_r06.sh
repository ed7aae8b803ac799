# rocprofv3 passes behind profiles/r06_* (as in round 5: the counter passes leave out bench.py's dense-equivalent leg -- WS_BENCH_SKIP_DENSE_EQ=1 --
# so every kernel name in a pass belongs to one route; a third SQ group holds the lane counters) (run on the GPU box via gpurun); outputs under gpurun_out/, summaries are copied
# into profiles/ afterwards (tools/make_traffic.py writes profiles/pmc_traffic.json).
#   bash tools/profile_r06.sh [TAG] [quick]
# Counter passes are separate runs with --kernel-trace only (no --stats, no other trace domain), one TCC counter per run
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), the SQ counters in two groups of eight.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the traced runs leave out the multi-rank sections of bench.py (RCCL start-up, HIP-graph capture and a child process under the
# tracer aborted rocprofv3 once); the untraced bench run below has them
export WS_BENCH_SKIP_SHARDED=1
TAG=${1:-r06}
QUICK=${2:-}
mkdir -p gpurun_out
for mode in sparse dense; do
  rm -rf gpurun_out/prof_${TAG}_${mode}
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${TAG}_${mode} -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --integrate ${mode} > gpurun_out/prof_${TAG}_${mode}.log 2>&1
  python tools/rocpd_stats.py $(ls gpurun_out/prof_${TAG}_${mode}/*.db gpurun_out/prof_${TAG}_${mode}/*/*.db 2>/dev/null | head -1) > gpurun_out/${TAG}_kernel_stats_${mode}.txt
done
if [ -z "$QUICK" ]; then
  WS_BENCH_SKIP_SHARDED=0 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
  for mode in sparse dense; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      WS_BENCH_SKIP_DENSE_EQ=1 rocprofv3 --kernel-trace --pmc ${ctr} -d gpurun_out/prof_${TAG}_pmc_${ctr}_${mode} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration --integrate ${mode} > gpurun_out/prof_${TAG}_pmc_${ctr}_${mode}.log 2>&1
      python tools/pmc_summary.py $(ls gpurun_out/prof_${TAG}_pmc_${ctr}_${mode}/*.db gpurun_out/prof_${TAG}_pmc_${ctr}_${mode}/*/*.db 2>/dev/null | head -1) > gpurun_out/${TAG}_pmc_${ctr}_${mode}.txt
    done
  done
  # where the cycles of the march / resolve kernels go (the issue-bound claim needs tracked evidence)
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    WS_BENCH_SKIP_DENSE_EQ=1 rocprofv3 --kernel-trace --pmc ${grp} -d gpurun_out/prof_${TAG}_pmc_sq${i} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration > gpurun_out/prof_${TAG}_pmc_sq${i}.log 2>&1
    python tools/pmc_summary.py $(ls gpurun_out/prof_${TAG}_pmc_sq${i}/*.db gpurun_out/prof_${TAG}_pmc_sq${i}/*/*.db 2>/dev/null | head -1) | grep -E "^kernel|march|resolve|ray_s" > gpurun_out/${TAG}_pmc_valu_sq${i}.txt
  done
  bash tools/pmc_lanes.sh ${TAG} > /dev/null 2>&1
  bash tools/read_calib.sh ${TAG} > /dev/null 2>&1   # FETCH_SIZE per byte loaded, by load shape (round 6)
  [ -x tools/valu_rate.out ] && ./tools/valu_rate.out > gpurun_out/${TAG}_valu_rate.txt 2>&1
  python tools/make_valu.py gpurun_out/${TAG}_pmc_valu_sq1.txt > gpurun_out/${TAG}_valu.log 2>&1 && cp profiles/pmc_valu.json gpurun_out/${TAG}_pmc_valu.json
  python tools/make_traffic.py gpurun_out/prof_${TAG} > gpurun_out/${TAG}_traffic.log 2>&1
  cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
fi
head -20 gpurun_out/${TAG}_kernel_stats_sparse.txt
