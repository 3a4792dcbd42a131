#!/bin/bash
# usage: tools/build_wt_variant.sh <name> "<extra compiler flags>"  -- build the WORKING TREE's old-vpic_amd/csrc with extra
# flags (e.g. -DVPIC_HIP_DEPOSIT_BLOCK=16) into tools/ab/lib<name>.so, for A/B timing with tools/ab.sh on one GPU box
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/old-vpic_amd/csrc $tmp/include $root/tools/ab
cp $root/old-vpic_amd/csrc/*.hip $root/old-vpic_amd/csrc/*.h $root/old-vpic_amd/csrc/Makefile $tmp/old-vpic_amd/csrc/
cp $root/include/*.h $tmp/include/
make -s -C $tmp/old-vpic_amd/csrc -j4 CXXEXTRA="$2"
cp $tmp/old-vpic_amd/libvpic_hip.so $root/tools/ab/lib$1.so
rm -rf $tmp
echo built tools/ab/lib$1.so with "$2"
