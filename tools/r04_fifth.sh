#!/bin/bash
# fifth GPU pass of round 4: the whole GPU suite; staged positions with whole-span cell stores on / off (configs[3] slab);
# sort_interval 40 / 30 with the early sort; the bench line
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-600
show='import json,sys
j=json.loads(sys.stdin.readline())
print("  %.2f G pushes/s  %.2f ms/step  advance_p %.3f ms/launch  roofline %.3f" % (j["value"]/1e9, j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]))
for s in j.get("advance_p_by_species") or []: print("     species %d charged %s: %.3f ms/launch  frac %.3f" % (s["species"], s["charged"], s["avg_launch_ms"], s["frac"]))'
for rep in 1 2; do for st in 0 1; do
  echo "-- configs[3] slab, VPIC_HIP_STAGE=$st"
  VPIC_HIP_STAGE=$st timeout -k 10 200 python bench.py --no-cpu-baseline --deck trecon --sort-interval -20 --steps 20 --warmup 10 2>>$O/bench.err | tail -1 | python -c "$show"
done; done
for si in 40 30; do
echo "-- sort_interval $si"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --sort-interval $si --steps 80 --warmup 5 2>$O/si$si.err | tail -1 | python -c "$show"
done
echo "-- the bench line"
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench_full.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-300
