#!/bin/bash
# seventh GPU pass of round 4: the whole GPU suite; deterministic accumulation with integer run sums; one or two rounds per
# staged batch (configs[3] slab); sort_interval 40 / 30 through bench.py (the host stays two steps ahead at most)
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-600
show='import json,sys
j=json.loads(sys.stdin.readline())
print("  %.2f G pushes/s  %.2f ms/step  advance_p %.3f ms/launch  roofline %.3f" % (j["value"]/1e9, j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]))
for s in j.get("advance_p_by_species") or []: print("     species %d charged %s: %.3f ms/launch  frac %.3f" % (s["species"], s["charged"], s["avg_launch_ms"], s["frac"]))
c=j.get("check") or {}
print("     check: conserved %s drift %s early sorts %s" % (c.get("particles_conserved"), c.get("total_energy_drift"), [s.get("early_sorts") for s in c.get("species", [])]))'
echo "-- deterministic accumulation, default deck"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --accumulation deterministic --steps 10 --warmup 3 2>$O/det.err | tail -1 | python -c "$show"
echo "-- float accumulation, default deck"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --steps 10 --warmup 3 2>$O/flt.err | tail -1 | python -c "$show"
for rep in 1 2; do for lib in cur rounds2; do
  echo "-- configs[3] slab, $lib"
  if [ $lib = cur ]; then unset VPIC_HIP_LIB; else export VPIC_HIP_LIB=$PWD/tools/ab/lib$lib.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --deck trecon --sort-interval -20 --steps 20 --warmup 10 2>>$O/bench.err | tail -1 | python -c "$show"
done; done
unset VPIC_HIP_LIB
for si in 40 30; do
echo "-- sort_interval $si"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --sort-interval $si --steps 80 --warmup 5 2>$O/si$si.err | tail -1 | python -c "$show"
done
