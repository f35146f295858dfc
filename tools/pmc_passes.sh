#!/bin/bash
# PMC passes over one short bench run each (run on the GPU box: tools/pmc_passes.sh <tag>).
# Each pass is its own rocprofv3 run with counters only (no sys/hip/hsa tracing), as the pool requires.
set -e
TAG=${1:-v5}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_${TAG}_$name -o $name -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1
    python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/pmc_${TAG}_$name/*.db $R/gpurun_out/pmc_${TAG}_$name/*/*.db 2>/dev/null | head -1) > $R/gpurun_out/pmc_${TAG}_$name.txt
}
pass fl SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU SQ_BUSY_CYCLES
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES
pass rd FETCH_SIZE
pass wr WRITE_SIZE GRBM_GUI_ACTIVE
