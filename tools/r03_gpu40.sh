#!/bin/bash
# the following window on decks whose particles do NOT move together (hot species, adaptive sorting): does the sampling cost anything?
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
echo "== hot vth 0.6"; bash tools/ab.sh "nofollow cur" "--config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 10"
echo "== trecon"; bash tools/ab.sh "nofollow cur" "--deck trecon --sort-interval -20 --steps 40 --warmup 10"
echo "== sheet"; bash tools/ab.sh "nofollow cur" "--deck sheet --sort-interval -20 --steps 40 --warmup 10"
echo "== drift 512"; bash tools/ab.sh "nofollow cur" "--deck drift --grid 128 128 128 --ppc 512"
echo "== config 2 adaptive"; bash tools/ab.sh "nofollow cur" "--sort-interval -20 --steps 40 --warmup 10"
