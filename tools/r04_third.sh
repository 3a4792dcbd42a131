#!/bin/bash
# third GPU pass of round 4: the whole GPU suite, kernel stats of the default deck (the LDS-tiled clear_jf + unload against
# the per-voxel one), configs[1] as a deck on the C++ host, sort_interval 40 with the early sort
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-400
echo "-- kernel stats, default deck"
timeout -k 10 300 bash tools/prof_stats.sh r04c > $O/prof_stats.txt 2>&1; tail -16 $O/prof_stats.txt; cp gpurun_out/r04c_kernel_stats.csv $O/ 2>/dev/null
echo "-- the same with VPIC_HIP_UNLOAD_TILED=0"
VPIC_HIP_UNLOAD_TILED=0 timeout -k 10 300 bash tools/prof_stats.sh r04c_untiled > $O/prof_stats_untiled.txt 2>&1; grep -i "unload" $O/prof_stats_untiled.txt
rm -rf gpurun_out/prof_r04c gpurun_out/prof_r04c_untiled
echo "-- configs[1] as a deck on the C++ host (adaptive sorting = the host's default, then fixed intervals)"
mkdir -p $O/deck && cd $O/deck
VPIC_HIP_HOST_TIMING=1 timeout -k 10 300 ../../../old-vpic_amd/host/twostream128.hip.exe -tpp=1 40 2>&1 | grep -i "simulation time\|hip host\|rror"
VPIC_HIP_ADAPTIVE_SORT=0 VPIC_HIP_HOST_TIMING=1 timeout -k 10 300 ../../../old-vpic_amd/host/twostream128.hip.exe -tpp=1 40 2>&1 | grep -i "simulation time\|hip host\|rror"
cd ../../..
echo "-- sort_interval 40 (early sorts when the deposits miss the windows)"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --sort-interval 40 --steps 80 --warmup 5 2>$O/si40.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('sort_interval 40: %.2f G pushes/s  %.2f ms/step  advance_p %.2f ms/launch  roofline %.3f check %s' % (j['value']/1e9, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j.get('check')))"
