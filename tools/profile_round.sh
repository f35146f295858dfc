#!/bin/bash
# Round profile on the GPU box: kernel trace + separate PMC passes (counters only, one rocprofv3 run each).
# usage: tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>_*.txt
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 args...
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 "$@" -d $R/gpurun_out/prof_${TAG}_$name -o $name -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-brute-force-leg > $R/gpurun_out/prof_${TAG}_$name.log 2>&1 || return 1
    python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/prof_${TAG}_$name/*.db $R/gpurun_out/prof_${TAG}_$name/*/*.db 2>/dev/null | head -1) > $R/gpurun_out/prof_${TAG}_$name.txt
}
run trace --kernel-trace --stats &&
run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE &&
run rd --kernel-trace --pmc FETCH_SIZE &&
run wr --kernel-trace --pmc WRITE_SIZE &&
run lds --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES
