#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
echo "== hot vth 0.6"; bash tools/ab.sh "nofollow cur" "--config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 10"
echo "== hot vth 0.24"; bash tools/ab.sh "nofollow cur" "--config 1 --vth 0.24 --sort-interval -20 --steps 40 --warmup 10"
echo "== config 2, interval 20"; bash tools/ab.sh "nofollow cur" "--steps 40 --warmup 5 --sort-interval 20"
echo "== config 1, interval 20"; bash tools/ab.sh "nofollow cur" "--config 1 --steps 40 --warmup 5 --sort-interval 20"
