#!/bin/bash
# the last pass of round 4 on the final tree: smoke, the bench line with its wall clock, the GPU suite
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04z; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s); python bench.py > $O/r04_bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -1 $O/r04_bench.json | cut -c1-300
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-300
