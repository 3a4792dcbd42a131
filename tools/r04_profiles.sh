#!/bin/bash
# the measurements behind profiles/r04_* and DESIGN.md section 5 (one GPU box):  gpurun -- bash tools/r04_profiles.sh
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04p; mkdir -p $O
timeout -k 10 700 python bench.py > $O/r04_bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/r04_bench.json | cut -c1-300
echo "-- kernel stats"
timeout -k 10 300 bash tools/prof_stats.sh r04 > $O/prof_stats.txt 2>&1 && cp gpurun_out/r04_kernel_stats.csv $O/r04_bench_kernel_stats.csv && tail -18 $O/prof_stats.txt
timeout -k 10 200 bash tools/prof_stats.sh r04c1 --config 1 > $O/prof_stats_c1.txt 2>&1 && cp gpurun_out/r04c1_kernel_stats.csv $O/r04_config1_kernel_stats.csv
timeout -k 10 200 bash tools/prof_stats.sh r04trecon --deck trecon --sort-interval -20 --steps 20 --warmup 10 > $O/prof_stats_trecon.txt 2>&1 && cp gpurun_out/r04trecon_kernel_stats.csv $O/r04_config3_slab_kernel_stats.csv && tail -14 $O/prof_stats_trecon.txt
echo "-- traffic"
timeout -k 10 400 bash tools/pmc_traffic.sh r04c2 --no-second-config > $O/pmc_c2.txt 2>&1 && cp gpurun_out/traffic_r04c2_raw.json $O/r04_traffic_config2_raw.json
timeout -k 10 300 bash tools/pmc_traffic.sh r04c1 --no-second-config --config 1 > $O/pmc_c1.txt 2>&1 && cp gpurun_out/traffic_r04c1_raw.json $O/r04_traffic_config1_raw.json
timeout -k 10 300 bash tools/pmc_traffic.sh r04c3 --deck trecon --sort-interval -20 --steps 20 --warmup 10 > $O/pmc_c3.txt 2>&1 && cp gpurun_out/traffic_r04c3_raw.json $O/r04_traffic_config3_slab_raw.json
tail -6 $O/pmc_c2.txt $O/pmc_c1.txt $O/pmc_c3.txt | cut -c1-400
rm -rf gpurun_out/pmc_r04c2_* gpurun_out/pmc_r04c1_* gpurun_out/pmc_r04c3_* gpurun_out/prof_r04 gpurun_out/prof_r04c1 gpurun_out/prof_r04trecon
echo "-- other decks and modes"
show='import json,sys
j=json.loads(sys.stdin.readline())
print("  %.2f G pushes/s  %.2f ms/step  advance_p %.3f ms/launch  roofline %.3f  %s" % (j["value"]/1e9, j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["config"]["workload"]))
for s in j.get("advance_p_by_species") or []: print("     species %d charged %s: %.3f ms/launch  frac %.3f" % (s["species"], s["charged"], s["avg_launch_ms"], s["frac"]))
c=j.get("check") or {}
print("     check: conserved %s drift %s early sorts %s" % (c.get("particles_conserved"), c.get("total_energy_drift"), [s.get("early_sorts") for s in c.get("species", [])]))'
timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-config --accumulation deterministic --steps 10 --warmup 3 2>>$O/bench.err > $O/r04_bench_deterministic.json; tail -1 $O/r04_bench_deterministic.json | python -c "$show"
timeout -k 10 200 python bench.py --no-cpu-baseline --config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 20 2>>$O/bench.err > $O/r04_bench_hot_vth06_adaptive.json; tail -1 $O/r04_bench_hot_vth06_adaptive.json | python -c "$show"
timeout -k 10 200 python bench.py --no-cpu-baseline --deck trecon --sort-interval -20 --steps 40 --warmup 20 2>>$O/bench.err > $O/r04_bench_config3_slab_adaptive.json; tail -1 $O/r04_bench_config3_slab_adaptive.json | python -c "$show"
VPIC_HIP_STAGE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --deck trecon --sort-interval -20 --steps 40 --warmup 20 2>>$O/bench.err > $O/r04_bench_config3_slab_positions_not_staged.json; tail -1 $O/r04_bench_config3_slab_positions_not_staged.json | python -c "$show"
echo "-- sort decisions: intervals, the heated phase (tools/early_sort_check.sh)"
bash tools/early_sort_check.sh > $O/early_sort_check.txt 2>&1; cp gpurun_out/r04es/r04_sort_interval_sweep.txt $O/r04_sort_interval_sweep.txt
(cat gpurun_out/r04es/r04_sustained.txt; grep -v amdgpu.ids gpurun_out/r04es/r04_heated_phase_step_by_step.txt | tail -34) > $O/r04_heated_phase_steps_160_200.txt; head -14 $O/early_sort_check.txt
echo "-- one-launch ablation (tools/ablate_once.py)"
VPIC_HIP_LIB=$PWD/tools/ab/libablation.so timeout -k 10 300 python tools/ablate_once.py 0 256 512 32 288 64 2 > $O/r04_ablate_once_config2.txt 2>&1; cat $O/r04_ablate_once_config2.txt
VPIC_HIP_LIB=$PWD/tools/ab/libablation.so timeout -k 10 300 python tools/ablate_once.py --deck trecon --steps-before 8 0 256 512 32 288 64 2 > $O/r04_ablate_once_config3_slab.txt 2>&1; cat $O/r04_ablate_once_config3_slab.txt
