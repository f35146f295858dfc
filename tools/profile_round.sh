#!/bin/bash
# Round profile on the GPU box: kernel trace + separate PMC passes (counters only, one rocprofv3 run each), then the
# JSON bench.py reads its instruction counts, clock and HBM traffic from.
# usage: tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>_*.txt, gpurun_out/prof_<tag>_pmc.json
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs"
cd /tmp && export TMPDIR=/tmp
db() { ls $R/gpurun_out/prof_${TAG}_$1/*.db $R/gpurun_out/prof_${TAG}_$1/*/*.db 2>/dev/null | head -1; }
run() {  # name, rocprofv3 args...
    local name=$1; shift
    timeout -k 10 -s KILL 200 rocprofv3 "$@" -d $R/gpurun_out/prof_${TAG}_$name -o $name -- $CMD > $R/gpurun_out/prof_${TAG}_$name.log 2>&1 || return 1
    python3 $R/tools/rocprof_summary.py $(db $name) > $R/gpurun_out/prof_${TAG}_$name.txt
}
run trace --kernel-trace --stats &&
run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE &&
run rd --kernel-trace --pmc FETCH_SIZE &&
run wr --kernel-trace --pmc WRITE_SIZE &&
run fl --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU SQ_INSTS_VALU_INT32 &&
run fl2 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_MFMA &&
python3 $R/tools/pmc_json.py $R/gpurun_out/prof_${TAG}_pmc.json synthetic-10000-1000-3840x2160-d4-spp1 \
    "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs" \
    $(db sq) $(db rd) $(db wr) $(db fl) $(db fl2) > /dev/null &&
run lds --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES
