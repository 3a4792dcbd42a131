#!/bin/bash
# the measurements behind profiles/r02_* and DESIGN.md section 5 (one GPU box):  tools/r02_profiles.sh
cd "$(dirname "$0")/.."; ulimit -c 0
O=gpurun_out/r02p; mkdir -p $O
python bench.py > $O/r02_bench.json 2> $O/bench.err && tail -1 $O/r02_bench.json | cut -c1-300
# kernel statistics of the same command (rocprofv3 --kernel-trace --stats)
bash tools/prof_stats.sh r02 > $O/prof_stats.txt 2>&1 && cp gpurun_out/r02_kernel_stats.csv $O/r02_bench_kernel_stats.csv && tail -18 $O/prof_stats.txt
# HBM traffic of advance_p_kernel (TCC counters, separate passes), both decks, exact arithmetic
bash tools/pmc_traffic.sh r02c2 --no-second-config > $O/pmc_c2.txt 2>&1 && cp gpurun_out/traffic_r02c2_raw.json $O/r02_traffic_config2_raw.json
bash tools/pmc_traffic.sh r02c1 --no-second-config --config 1 > $O/pmc_c1.txt 2>&1 && cp gpurun_out/traffic_r02c1_raw.json $O/r02_traffic_config1_raw.json
tail -4 $O/pmc_c2.txt $O/pmc_c1.txt
rm -rf gpurun_out/pmc_r02c2_* gpurun_out/pmc_r02c1_* gpurun_out/prof_r02
# SQ counters of advance_p (VALU instructions per 64 particles and the LDS share), config 1 and 2
S="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;GRBM_GUI_ACTIVE"
bash tools/pmc_sets.sh r02sq_c2_exact "$S" --steps 10 --warmup 3 > $O/r02_sq_config2_exact.txt 2>&1
bash tools/pmc_sets.sh r02sq_c2_fast "$S" --steps 10 --warmup 3 --push fast > $O/r02_sq_config2_fast.txt 2>&1
bash tools/pmc_sets.sh r02sq_c1_exact "$S" --config 1 --steps 10 --warmup 3 > $O/r02_sq_config1_exact.txt 2>&1
# other decks
python bench.py --no-cpu-baseline --config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 20 > $O/r02_bench_hot_vth06_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 20 --push fast > $O/r02_bench_hot_vth06_adaptive_fast.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --config 1 --vth 0.24 --sort-interval -20 --steps 40 --warmup 20 > $O/r02_bench_hot_vth024_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --deck drift --grid 128 128 128 --ppc 512 --steps 10 --warmup 3 > $O/r02_bench_drift512.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --deck sheet --sort-interval -20 --steps 40 --warmup 20 > $O/r02_bench_sheet_adaptive.json 2>> $O/bench.err
for f in $O/r02_bench*.json; do echo $f; tail -1 $f | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('  %.3e pushes/s  %.3f ms/step  roofline %.3f  %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['workload']))"; done
