#!/bin/bash
# rocprofv3 kernel statistics of the production deck executable (oracle/_ref/treconbig.hip.exe, see time_trecon.sh)
cd "$(dirname "$0")/.."
python -c "import importlib; importlib.import_module('old-vpic_amd').lib()"
OUT=$PWD/gpurun_out/trecon_prof; rm -rf $OUT; mkdir -p $OUT/run
cd $OUT/run
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- /root/repo/oracle/_ref/treconbig.hip.exe -tpp=1 > log 2>&1 || true
grep -E "simulation time|rror" log
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cd $OUT && rm -rf run prof
