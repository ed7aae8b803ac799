# quick counter passes for the lib at HEAD (sparse mode only): SQ group 1, FETCH_SIZE, WRITE_SIZE; outputs gpurun_out/q_*.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WS_BENCH_SKIP_SHARDED=1
mkdir -p gpurun_out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf gpurun_out/prof_q_$tag
  rocprofv3 --kernel-trace --pmc ${grp} -d gpurun_out/prof_q_$tag -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration > gpurun_out/prof_q_$tag.log 2>&1
  python tools/pmc_summary.py $(ls gpurun_out/prof_q_$tag/*.db gpurun_out/prof_q_$tag/*/*.db 2>/dev/null | head -1) | grep -E "^kernel|march|resolve_kernel<false, true|ray_s" > gpurun_out/q_$tag.txt
  rm -rf gpurun_out/prof_q_$tag
done
cat gpurun_out/q_*.txt | cut -c1-200
