#!/bin/bash
# first GPU pass of round 4: the GPU tests (with the new bench-size parity case), the bench line of this box, and the ledger
# of the cell-crossing path by ablation (a -DVPIC_HIP_ABLATION build: tools/ab/libablation.so) on the cold headline deck and
# on the configs[3] slab.   usage: gpurun -- bash tools/r04_first.sh
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-400
export VPIC_HIP_LIB=$PWD/tools/ab/libablation.so
echo "-- ablation, 256^3 x 64 ppc (0 all on; 256 no late stores of the crossers' final positions; 512 one of the four; 32 no mover deposit; 64 no drain)"
timeout -k 10 500 bash tools/ablate.sh "0 256 512 32 288 64" > $O/ablate_cold.txt 2>&1; cat $O/ablate_cold.txt
echo "-- ablation, configs[3] slab (hot)"
timeout -k 10 400 bash tools/ablate.sh "0 256 512 32 288 64" "--deck trecon --sort-interval -20 --steps 20 --warmup 10" > $O/ablate_hot.txt 2>&1; cat $O/ablate_hot.txt
