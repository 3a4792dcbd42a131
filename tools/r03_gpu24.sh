#!/bin/bash
# XCD-contiguous chunks for the tile-only (coarse) scatter on hot decks; full-suite check of the tree
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03y; mkdir -p $O
echo "== hot cur"; bash tools/prof_stats.sh r03y_hot_cur --steps 40 --warmup 10 --config 1 --vth 0.6 --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
echo "== hot cxcd"; VPIC_HIP_LIB=$PWD/tools/ab/libcxcd.so bash tools/prof_stats.sh r03y_hot_cxcd --steps 40 --warmup 10 --config 1 --vth 0.6 --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
echo "== trecon cur"; bash tools/prof_stats.sh r03y_tr_cur --steps 40 --warmup 10 --deck trecon --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
echo "== trecon cxcd"; VPIC_HIP_LIB=$PWD/tools/ab/libcxcd.so bash tools/prof_stats.sh r03y_tr_cxcd --steps 40 --warmup 10 --deck trecon --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
