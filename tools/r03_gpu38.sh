#!/bin/bash
# the push kernels' VGPR budget (80: set when six workgroups per CU were in reach; the LDS allows five, which 102 VGPRs still fit): 96, 100
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
echo "== config 2"; bash tools/ab.sh "cur v96 v100" ""
echo "== config 1"; bash tools/ab.sh "cur v96 v100" "--config 1"
echo "== hot"; bash tools/ab.sh "cur v96 v100" "--config 1 --vth 0.6 --sort-interval -20 --steps 30 --warmup 10"
