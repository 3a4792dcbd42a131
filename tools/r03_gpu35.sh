#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tiles.py -m gpu -q -k "inside_the_push" > $O/pytest1.log 2>&1; echo "pytest(fuse) rc=$?"; tail -2 $O/pytest1.log
bash tools/ab.sh "prev cur" "--deck drift --grid 128 128 128 --ppc 512"
bash tools/ab.sh "prev cur" ""
bash tools/ab.sh "prev cur" "--config 1"
