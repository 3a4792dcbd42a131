#!/bin/bash
# build and run oracle/decks/sheet4.cxx on the HIP host, keep its outputs under gpurun_out/sheet4[_n2]
#   tools/run_sheet4.sh [nranks]
set -e
cd "$(dirname "$0")/.."
N=${1:-1}
python -c "import importlib; importlib.import_module('old-vpic_amd').lib()"
OUT=gpurun_out/sheet4; [ "$N" != 1 ] && OUT=gpurun_out/sheet4_n$N
rm -rf $OUT && mkdir -p $OUT
if [ "$N" = 1 ]; then
  make -s -C old-vpic_amd/host deck DECK=$PWD/oracle/decks/sheet4.cxx OUT=$PWD/$OUT/sheet4
  (cd $OUT && ./sheet4.hip.exe -tpp=1 > log 2>&1)
else
  make -s -C old-vpic_amd/host deck MPI=1 DECK=$PWD/oracle/decks/sheet4.cxx OUT=$PWD/$OUT/sheet4
  (cd $OUT && timeout -k 10 300 /opt/conda/bin/mpiexec -n $N ./sheet4.hip.exe -tpp=1 > log 2>&1)
fi
rm -f $OUT/sheet4.hip.exe $OUT/rundata/grid.*
