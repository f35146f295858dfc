#!/bin/bash
# A/B of kernel build variants: runs bench.py once per library in raytrace_amd/lib/ab/ (same process
# conditions, back to back) and prints value / kernel times.  usage: tools/ab_bench.sh [bench args]
for lib in raytrace_amd/lib/ab/*.so; do
  out=$(RT_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1) || { echo "$lib FAILED"; exit 1; }
  echo "$(basename $lib) $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["config"]["kernel_ms"]; print("%.1f Mrays/s  frame %.2f ms  primary %.2f  setup %.2f  shadow %.2f  final %.2f" % (d["value"], d["ms_per_step"], k["k_primary"], k["k_setup"], k["k_shadow"], k["k_final"]))')"
done
