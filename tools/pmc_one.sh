#!/bin/bash
# One PMC pass: tools/pmc_one.sh <name> <counter>...   (run on the GPU box; counters only, no tracing domains)
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_$name -o $name -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-brute-force-leg > $R/gpurun_out/pmc_$name.log 2>&1
rc=$?
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/pmc_$name/*.db $R/gpurun_out/pmc_$name/*/*.db 2>/dev/null | head -1) > $R/gpurun_out/pmc_$name.txt
exit $rc
