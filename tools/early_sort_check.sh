#!/bin/bash
# the sort decisions of vpic_hip_step (early sorts, sort inside the push or before it): the sort-interval sweep and the heated
# phase (steps 160..200) of the headline deck, then that phase step by step
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04es; mkdir -p $O
show='import json,sys
j=json.loads(sys.stdin.readline())
print("  %.2f G pushes/s  %.2f ms/step  advance_p %.3f ms/launch  roofline %.3f  %s" % (j["value"]/1e9, j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["config"]["workload"]))
c=j.get("check") or {}
print("     check: conserved %s drift %s sorts %s early sorts %s; launches that sorted as they pushed: %s" % (c.get("particles_conserved"), c.get("total_energy_drift"), [s.get("sorts") for s in c.get("species", [])], [s.get("early_sorts") for s in c.get("species", [])], (j.get("advance_p_sorting") or {}).get("launches")))'
for si in 10 20 30 40 60; do echo -n "sort_interval $si: "; timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --sort-interval $si --steps $((si > 30 ? 2*si : 60)) --warmup 5 2>>$O/bench.err | tail -1 | python -c "$show"; done 2>&1 | tee $O/r04_sort_interval_sweep.txt
(echo -n "steps 160..200, sort_interval 10: "; timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-config --steps 40 --warmup 160 2>>$O/bench.err | tail -1 | python -c "$show") 2>&1 | tee $O/r04_sustained.txt
timeout -k 10 300 python tools/si_trace.py 10 200 168 > $O/r04_heated_phase_step_by_step.txt 2>&1; tail -32 $O/r04_heated_phase_step_by_step.txt | cut -c1-150
