#!/bin/bash
# sixth GPU pass of round 4: the whole GPU suite; sort_interval 40 watched step by step
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | cut -c1-600
echo "-- sort_interval 40, step by step"
timeout -k 10 300 python tools/si_trace.py 40 90 > $O/si40_trace.txt 2>&1; cat $O/si40_trace.txt | cut -c1-300
