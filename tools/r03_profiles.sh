#!/bin/bash
# the measurements behind profiles/r03_* and DESIGN.md section 5 (one GPU box):  tools/r03_profiles.sh
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03p; mkdir -p $O
python bench.py > $O/r03_bench.json 2> $O/bench.err && tail -1 $O/r03_bench.json | cut -c1-300
echo "-- kernel stats"
bash tools/prof_stats.sh r03 > $O/prof_stats.txt 2>&1 && cp gpurun_out/r03_kernel_stats.csv $O/r03_bench_kernel_stats.csv && tail -18 $O/prof_stats.txt
bash tools/prof_stats.sh r03c1 --config 1 > $O/prof_stats_c1.txt 2>&1 && cp gpurun_out/r03c1_kernel_stats.csv $O/r03_config1_kernel_stats.csv
bash tools/prof_stats.sh r03trecon --deck trecon --sort-interval -20 --steps 20 --warmup 10 > $O/prof_stats_trecon.txt 2>&1 && cp gpurun_out/r03trecon_kernel_stats.csv $O/r03_config3_slab_kernel_stats.csv && tail -14 $O/prof_stats_trecon.txt
echo "-- traffic"
bash tools/pmc_traffic.sh r03c2 --no-second-config > $O/pmc_c2.txt 2>&1 && cp gpurun_out/traffic_r03c2_raw.json $O/r03_traffic_config2_raw.json
bash tools/pmc_traffic.sh r03c1 --no-second-config --config 1 > $O/pmc_c1.txt 2>&1 && cp gpurun_out/traffic_r03c1_raw.json $O/r03_traffic_config1_raw.json
tail -4 $O/pmc_c2.txt $O/pmc_c1.txt
rm -rf gpurun_out/pmc_r03c2_* gpurun_out/pmc_r03c1_* gpurun_out/prof_r03 gpurun_out/prof_r03c1 gpurun_out/prof_r03trecon
echo "-- SQ counters"
S="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;GRBM_GUI_ACTIVE"
KERNEL="2, false, false>" bash tools/pmc_sets.sh r03sq_c2_exact "$S" --steps 10 --warmup 3 > $O/r03_sq_config2_exact.txt 2>&1
KERNEL="2, false, false>" bash tools/pmc_sets.sh r03sq_c1_exact "$S" --config 1 --steps 10 --warmup 3 > $O/r03_sq_config1_exact.txt 2>&1
KERNEL="2, false, true>" bash tools/pmc_sets.sh r03sq_c2_sorting "$S" --steps 12 --warmup 9 > $O/r03_sq_config2_sorting_push.txt 2>&1
tail -20 $O/r03_sq_config2_exact.txt
echo "-- other decks"
python bench.py --no-cpu-baseline --config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 20 > $O/r03_bench_hot_vth06_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --config 1 --vth 0.24 --sort-interval -20 --steps 40 --warmup 20 > $O/r03_bench_hot_vth024_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --deck drift --grid 128 128 128 --ppc 512 --steps 10 --warmup 3 > $O/r03_bench_drift512.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --deck sheet --sort-interval -20 --steps 40 --warmup 20 > $O/r03_bench_sheet_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --deck trecon --sort-interval -20 --steps 40 --warmup 20 > $O/r03_bench_config3_slab_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-second-config --accumulation deterministic --steps 10 --warmup 3 > $O/r03_bench_deterministic.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-second-config --sort-interval -20 --steps 40 --warmup 10 > $O/r03_bench_adaptive.json 2>> $O/bench.err
VPIC_HIP_SORT_IN_PUSH=0 python bench.py --no-cpu-baseline --no-second-config > $O/r03_bench_sort_then_push.json 2>> $O/bench.err
echo "-- sort intervals (the window follows the particles between sorts)"
for si in 10 15 20 30 40; do python bench.py --no-cpu-baseline --no-second-config --sort-interval $si --steps $((2*si)) --warmup 5 2>> $O/bench.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('sort_interval $si: %.2f G pushes/s  %.2f ms/step  advance_p %.2f ms/launch  roofline %.3f' % (j['value']/1e9, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac']))"; done | tee $O/r03_sort_interval_sweep.txt
for f in $O/r03_bench*.json; do echo $f; tail -1 $f | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('  %.3e pushes/s  %.3f ms/step  roofline %.3f  %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['workload']))"; done
