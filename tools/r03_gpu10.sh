#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03j; mkdir -p $O
echo "== config 2 (256^3, 64 ppc)"; bash tools/ab.sh "cur mq64 mq96 mq64t512" "" 2>&1 | tee $O/ab_c2.txt
echo "== config 1 (128^3, 32 ppc)"; bash tools/ab.sh "cur mq64 mq96 mq64t512" "--config 1" 2>&1 | tee $O/ab_c1.txt
echo "== hot"; bash tools/ab.sh "cur mq64 mq64t512" "--deck trecon --sort-interval -20" 2>&1 | tee $O/ab_hot.txt
