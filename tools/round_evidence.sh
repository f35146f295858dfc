# Round evidence on the GPU box (everything under profiles/<tag>_* except the rocprofv3 passes: tools/profile_round.sh).
# Build the timeline variant first, here on the CPU: tools/build_variant.sh wt -DRT_WAVE_TIMES
# usage: tools/round_evidence.sh [tag=r03]      (chain: every step under its own time limit, the first failure ends the script)
set -e
T=${1:-r03}
O=gpurun_out/ev
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -v --durations=12 --timeout 120 --timeout-method thread > $O/${T}_gpu_tests.txt 2>&1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.txt 2>&1
timeout -k 10 300 python bench.py > $O/${T}_bench.json 2> $O/bench.err
timeout -k 10 120 python bench.py --workload mirror --no-cpu-baseline --no-extra-legs > $O/${T}_bench_mirror.json 2>> $O/bench.err
timeout -k 10 120 python bench.py --workload default --no-cpu-baseline --no-extra-legs > $O/${T}_bench_default_scene_4k.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --workload cylinders --width 7680 --height 4320 --spp 16 --steps 3 --no-cpu-baseline --no-extra-legs > $O/${T}_bench_cylinders_8k_spp16.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --filter valu --no-cpu-baseline --no-extra-legs > $O/${T}_bench_valu_filter.json 2>> $O/bench.err
timeout -k 10 120 python tools/host_render_timing.py > $O/${T}_host_render_timing.txt 2>&1
timeout -k 10 120 python tools/bounce_stats.py > $O/${T}_bounce_levels.txt 2>&1
RT_LIB=raytrace_amd/lib/ab/wt.so timeout -k 10 120 python tools/wave_times.py > $O/${T}_wave_timeline.txt 2>&1
RT_LIB=raytrace_amd/lib/ab/wt.so timeout -k 10 120 python tools/wave_times.py mirror > $O/${T}_wave_timeline_k_general.txt 2>&1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --same-device --verify > $O/${T}_gloo_2rank_rehearsal.json 2> $O/gloo.err
echo ALL_OK
