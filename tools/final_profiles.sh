#!/bin/bash
# the measurements behind profiles/ and DESIGN.md section 5, one after the other (one GPU box):
#   tools/final_profiles.sh <round tag, e.g. r01>
cd "$(dirname "$0")/.."
T=${1:-r01}; O=gpurun_out/final; mkdir -p $O
python bench.py > $O/${T}_bench.json 2> $O/bench.err && tail -1 $O/${T}_bench.json | cut -c1-300
bash tools/prof_stats.sh $T > $O/prof_stats.txt 2>&1 && cp gpurun_out/${T}_kernel_stats.csv $O/${T}_bench_kernel_stats.csv && tail -16 $O/prof_stats.txt
bash tools/pmc_traffic.sh $T > $O/pmc_traffic.txt 2>&1 && cp gpurun_out/traffic_${T}_raw.json $O/${T}_traffic_raw.json && tail -8 $O/pmc_traffic.txt
python bench.py --no-cpu-baseline --ppc 64 > $O/${T}_bench_ppc64.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --deck drift --ppc 512 --steps 10 --warmup 3 > $O/${T}_bench_drift512.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --deck sheet --sort-interval -20 --steps 40 --warmup 20 > $O/${T}_bench_sheet_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --vth 0.6 --sort-interval -20 --steps 40 --warmup 20 > $O/${T}_bench_hot_vth06_adaptive.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --vth 0.24 --sort-interval -20 --steps 40 --warmup 20 > $O/${T}_bench_hot_vth024_adaptive.json 2>> $O/bench.err
for f in $O/${T}_bench_*.json; do echo $f; tail -1 $f | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('  %.3e pushes/s  %.3f ms/step  roofline %.3f  %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['workload']))"; done
