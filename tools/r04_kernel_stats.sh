#!/bin/bash
# rocprofv3 --kernel-trace --stats of the three bench decks (part of tools/r04_profiles.sh; by itself when only the summaries are wanted)
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04p; mkdir -p $O
timeout -k 10 300 bash tools/prof_stats.sh r04 > $O/prof_stats.txt 2>&1 && cp gpurun_out/r04_kernel_stats.csv $O/r04_bench_kernel_stats.csv && tail -18 $O/prof_stats.txt
timeout -k 10 200 bash tools/prof_stats.sh r04c1 --config 1 > $O/prof_stats_c1.txt 2>&1 && cp gpurun_out/r04c1_kernel_stats.csv $O/r04_config1_kernel_stats.csv
timeout -k 10 200 bash tools/prof_stats.sh r04trecon --deck trecon --sort-interval -20 --steps 20 --warmup 10 > $O/prof_stats_trecon.txt 2>&1 && cp gpurun_out/r04trecon_kernel_stats.csv $O/r04_config3_slab_kernel_stats.csv && tail -14 $O/prof_stats_trecon.txt
rm -rf gpurun_out/prof_r04 gpurun_out/prof_r04c1 gpurun_out/prof_r04trecon
