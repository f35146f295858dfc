# Round evidence on the GPU box (everything under profiles/r02_* except the rocprofv3 passes).  Build the timeline variant first,
# here on the CPU: tools/build_variant.sh wt -DRT_WAVE_TIMES
set -e
O=gpurun_out/ev
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -v --durations=12 --timeout 120 --timeout-method thread > $O/r02_gpu_tests.txt 2>&1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_smoke.txt 2>&1
timeout -k 10 300 python bench.py > $O/r02_bench.json 2> $O/bench.err
timeout -k 10 120 python bench.py --workload mirror --no-cpu-baseline --no-extra-legs > $O/r02_bench_mirror.json 2>> $O/bench.err
timeout -k 10 120 python bench.py --workload default --no-cpu-baseline --no-extra-legs > $O/r02_bench_default_scene_4k.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --workload cylinders --width 7680 --height 4320 --spp 16 --steps 3 --no-cpu-baseline --no-extra-legs > $O/r02_bench_cylinders_8k_spp16.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --filter valu --no-cpu-baseline --no-extra-legs > $O/r02_bench_valu_filter.json 2>> $O/bench.err
timeout -k 10 120 python tools/host_render_timing.py 1 2 > $O/r02_host_render_timing.txt 2>&1
timeout -k 10 120 python tools/bounce_stats.py > $O/r02_bounce_levels.txt 2>&1
RT_LIB=raytrace_amd/lib/ab/wt.so timeout -k 10 120 python tools/wave_times.py > $O/r02_wave_timeline.txt 2>&1
RT_LIB=raytrace_amd/lib/ab/wt.so timeout -k 10 120 python tools/wave_times.py mirror > $O/r02_wave_timeline_k_general.txt 2>&1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --dist-backend gloo --same-device --verify > $O/r02_gloo_2rank_rehearsal.json 2> $O/gloo.err
echo ALL_OK
