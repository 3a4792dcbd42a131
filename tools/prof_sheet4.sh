#!/bin/bash
# rocprofv3 kernel statistics of the sheet4 deck on the C++ host:  tools/prof_sheet4.sh [NX NY NZ PPC STEPS]
set -e
cd "$(dirname "$0")/.."
NX=${1:-128}; NY=${2:-64}; NZ=${3:-128}; PPC=${4:-32}; STEPS=${5:-50}
python -c "import importlib; importlib.import_module('old-vpic_amd').lib()"
OUT=$PWD/gpurun_out/sheet4_prof
rm -rf $OUT && mkdir -p $OUT
make -s -C old-vpic_amd/host deck DECK=$PWD/oracle/decks/sheet4.cxx OUT=$OUT/sheet4 \
  DECK_DEFS="-DSHEET_NX=$NX -DSHEET_NY=$NY -DSHEET_NZ=$NZ -DSHEET_PPC=$PPC -DSHEET_STEPS=$STEPS"
cd $OUT
export TMPDIR=/tmp VPIC_HIP_HOST_TIMING=1 VPIC_HIP_MIRROR_INTERVAL=$STEPS
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- ./sheet4.hip.exe -tpp=1 > log 2>&1 || true
grep -E "hip host timing|simulation time" log || true
find prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf prof sheet4.hip.exe fields hydro rundata *.bin
