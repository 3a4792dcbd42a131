#!/bin/bash
# fourth GPU pass of round 4: the whole GPU suite; the staged positions of the hot instances on / off (configs[3] slab, per
# species); kernel stats of the default deck (LDS-tiled clear_jf + unload with planes prefetched); sort_interval 40
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-600
show='import json,sys
j=json.loads(sys.stdin.readline())
print("  %.2f G pushes/s  %.2f ms/step  advance_p %.3f ms/launch  roofline %.3f" % (j["value"]/1e9, j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]))
for s in j.get("advance_p_by_species") or []: print("     species %d charged %s: %.3f ms/launch  frac %.3f" % (s["species"], s["charged"], s["avg_launch_ms"], s["frac"]))'
for rep in 1 2; do for st in 0 1; do
  echo "-- configs[3] slab, VPIC_HIP_STAGE=$st"
  VPIC_HIP_STAGE=$st timeout -k 10 200 python bench.py --no-cpu-baseline --deck trecon --sort-interval -20 --steps 20 --warmup 10 2>>$O/bench.err | tail -1 | python -c "$show"
done; done
echo "-- configs[3] slab, the engine's own switch"
timeout -k 10 200 python bench.py --no-cpu-baseline --deck trecon --sort-interval -20 --steps 20 --warmup 10 2>>$O/bench.err | tail -1 | python -c "$show"
echo "-- kernel stats, default deck"
timeout -k 10 300 bash tools/prof_stats.sh r04d > $O/prof_stats.txt 2>&1; tail -16 $O/prof_stats.txt; cp gpurun_out/r04d_kernel_stats.csv $O/ 2>/dev/null
rm -rf gpurun_out/prof_r04d
echo "-- sort_interval 40 (early sorts when the deposits miss the windows)"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --sort-interval 40 --steps 80 --warmup 5 2>$O/si40.err | tail -1 | python -c "$show"
