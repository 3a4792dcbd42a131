#!/bin/bash
# time oracle/decks/sheet4.cxx at a production-like size on the HIP host (C++ deck host, 1 rank):
#   tools/time_sheet4.sh [NX NY NZ PPC STEPS]      -> gpurun_out/sheet4_big/log (main.cxx prints "simulation time")
set -e
cd "$(dirname "$0")/.."
NX=${1:-128}; NY=${2:-64}; NZ=${3:-128}; PPC=${4:-32}; STEPS=${5:-100}
python -c "import importlib; importlib.import_module('old-vpic_amd').lib()"
OUT=gpurun_out/sheet4_big
rm -rf $OUT && mkdir -p $OUT
make -s -C old-vpic_amd/host deck DECK=$PWD/oracle/decks/sheet4.cxx OUT=$PWD/$OUT/sheet4 \
  DECK_DEFS="-DSHEET_NX=$NX -DSHEET_NY=$NY -DSHEET_NZ=$NZ -DSHEET_PPC=$PPC -DSHEET_STEPS=$STEPS"
cd $OUT
for ADAPT in 0 1; do
  VPIC_HIP_HOST_TIMING=1 VPIC_HIP_MIRROR_INTERVAL=$STEPS VPIC_HIP_ADAPTIVE_SORT=$ADAPT ./sheet4.hip.exe -tpp=1 > log_adapt$ADAPT 2>&1 || true
  echo "adaptive=$ADAPT"
  grep -E "simulation time|hip host timing|Error" log_adapt$ADAPT
  tail -1 energies4.txt
done
rm -rf sheet4.hip.exe fields hydro rundata *.bin
