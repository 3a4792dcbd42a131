#!/bin/bash
# time oracle/_ref/treconbig.hip.exe (the reference's production deck at 64x64x32 cells, 13 M particles + a tracer
# copy of each; built by: make -C oracle trecon TOPO=1 NAME=big EXTRA="-DVPIC_PARTICLE_X=64 -DVPIC_PARTICLE_Y=64
# -DVPIC_PARTICLE_Z=32 -DVPIC_TIMESTEPS=100 -DVPIC_DUMPS=1") on the GPU, and the reference executable beside it
cd "$(dirname "$0")/.."
python -c "import importlib; importlib.import_module('old-vpic_amd').lib()"
OUT=$PWD/gpurun_out/trecon_big; rm -rf $OUT; mkdir -p $OUT/hip $OUT/ref
(cd $OUT/hip && VPIC_HIP_HOST_TIMING=1 timeout -k 10 900 /root/repo/oracle/_ref/treconbig.hip.exe -tpp=1 > log 2>&1; grep -E "simulation time|hip host timing|rror" log; du -sh . | tail -1; rm -rf names particle hydro fields restart* tracer)
if [ "$1" = ref ]; then
  (cd $OUT/ref && timeout -k 10 900 /root/repo/oracle/_ref/treconbig.exe -tpp=1 > log 2>&1; grep -E "simulation time|rror" log; rm -rf names particle hydro fields restart* tracer)
fi
