#!/bin/bash
# build and run a variant of oracle/decks/plumbing16.cxx on the HIP host (1 rank), keep its outputs:
#   tools/run_deck_variant.sh "<DECK_DEFS>" <outdir under gpurun_out>
set -e
cd "$(dirname "$0")/.."
python -c "import importlib; importlib.import_module('old-vpic_amd').lib()"
OUT=gpurun_out/$2; rm -rf $OUT && mkdir -p $OUT
make -s -C old-vpic_amd/host deck DECK=$PWD/oracle/decks/plumbing16.cxx DECK_DEFS="$1" OUT=$PWD/$OUT/deck
(cd $OUT && ./deck.hip.exe -tpp=1 > log 2>&1)
rm -f $OUT/deck.hip.exe
