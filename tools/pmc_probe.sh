# diagnostic PMC passes for the TSDF kernels (separate runs, kernel-trace only); prints the march kernels' rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/probe_$i -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-registration > gpurun_out/probe_$i.log 2>&1
  python tools/pmc_summary.py gpurun_out/probe_$i/pmc_results.db 2>&1 | grep "march_kernel\|^kernel" | head -20
done
