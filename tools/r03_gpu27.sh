#!/bin/bash
# scan width of the main pass's deposits in the tile window (16 today): 4, 8, 64
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
echo "== config 2"; bash tools/ab.sh "cur mb4 mb8 mb64" ""
echo "== config 1"; bash tools/ab.sh "cur mb4 mb8 mb64" "--config 1"
