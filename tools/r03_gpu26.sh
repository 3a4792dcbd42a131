#!/bin/bash
# scan width of the crossers' deposits in the tile window (8 today): 1 (no scan), 2, 4 -- hot and cold decks
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
echo "== hot vth 0.6"; bash tools/ab.sh "cur db1 db2 db4" "--config 1 --vth 0.6 --sort-interval -20 --steps 30 --warmup 10"
echo "== trecon"; bash tools/ab.sh "cur db1 db2 db4" "--deck trecon --sort-interval -20 --steps 30 --warmup 10"
echo "== config 2"; bash tools/ab.sh "cur db1 db4" ""
