#!/bin/bash
# usage: tools/build_variant.sh <git-rev> <name>   -- build old-vpic_amd/csrc of <git-rev> into tools/ab/lib<name>.so
# (run a bench against it with VPIC_HIP_LIB=tools/ab/lib<name>.so; for A/B timing on one GPU box)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/old-vpic_amd/csrc $tmp/include $root/tools/ab
for f in $(git -C $root ls-tree --name-only $1 old-vpic_amd/csrc/ include/); do git -C $root show $1:$f > $tmp/$f; done
make -s -C $tmp/old-vpic_amd/csrc -j4
cp $tmp/old-vpic_amd/libvpic_hip.so $root/tools/ab/lib$2.so
rm -rf $tmp
echo built tools/ab/lib$2.so from $1
