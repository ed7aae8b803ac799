# diagnostic counter passes for the TSDF kernels (run on the GPU box via gpurun): bash tools/pmc_probe.sh TAG "CTR1 CTR2 ..."
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-probe}
CTRS=${2:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"}
rocprofv3 --kernel-trace --pmc ${CTRS} -d gpurun_out/pmc_${TAG} -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-registration ${3:-} > gpurun_out/pmc_${TAG}.log 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc_${TAG}/*.db gpurun_out/pmc_${TAG}/*/*.db 2>/dev/null | head -1) | grep -E "kernel|march|resolve|integrate|tile_|ray_|desc" | tee gpurun_out/pmc_${TAG}.txt
