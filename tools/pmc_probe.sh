# diagnostic PMC passes for the TSDF kernels (separate runs, kernel-trace only); summaries printed to stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/probe_$i -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-registration > gpurun_out/probe_$i.log 2>&1
  python tools/pmc_summary.py gpurun_out/probe_$i/pmc_results.db 2>&1 | grep -v "fill_u\|memset\|Memset" | head -60
done
