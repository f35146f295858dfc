#!/bin/bash
# Register / scratch / LDS use of every kernel of a build variant (compiles on the CPU, no GPU needed).
# usage: tools/kernel_resources.sh [-DFLAG=..]...
cd "$(dirname "$0")/../raytrace_amd/csrc"
mkdir -p /tmp/abbuild
for f in ${RT_KERNEL_SOURCES:-rt_kernels.hip}; do
/opt/rocm/bin/hipcc -O3 "$@" -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function \
    -Rpass-analysis=kernel-resource-usage -c -o /tmp/abbuild/res_$$.o $f 2>&1 |
python3 -c '
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m: cur = m.group(2); rows[cur] = {}; continue
    m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
print("%-28s %5s %5s %7s %4s %6s %6s %6s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "sSpill", "vSpill", "LDS"))
for k, r in rows.items():
    name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", k); name = re.sub(r"ENS_12RenderParamsE$|E[A-Z].*$", "", name)
    print("%-28s %5d %5d %7d %4d %6d %6d %6d" % (name[:28], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1), r.get("LDS Size", -1)))
'
done
rm -f /tmp/abbuild/res_$$.o
