#!/bin/bash
# the reference's production reconnection deck (decks/trecon-part/turbulence.cxx, unchanged) at the per-GPU size of BASELINE
# configs[3] -- 32 x 256 x 128 cells, 4 species x 50 ppc (the deck's own nppc) + a tracer copy of every particle -- on the C++ deck
# host: built where the reference tree is by
#   make -C oracle trecon TOPO=1 NAME=slab EXTRA="-DVPIC_PARTICLE_X=32 -DVPIC_PARTICLE_Y=256 -DVPIC_PARTICLE_Z=128 -DVPIC_TIMESTEPS=30 -DVPIC_DUMPS=1"
# (oracle/_ref/treconslab.hip.exe travels to the GPU box).  Prints the deck's own wall clock and the host's split of it.
cd "$(dirname "$0")/.."; ulimit -c 0
W=/tmp/trecon_slab_run; rm -rf $W; mkdir -p $W gpurun_out; cd $W
t0=$(date +%s)
( while sleep 50; do echo "  ... $(( $(date +%s) - t0 )) s"; done ) & KA=$!
VPIC_HIP_HOST_TIMING=1 timeout -k 10 1000 /root/repo/oracle/_ref/treconslab.hip.exe -tpp=1 > log 2>&1; rc=$?
kill $KA 2>/dev/null
echo "rc=$rc wall $(( $(date +%s) - t0 )) s"; grep -E "simulation time|hip host|rror|num_step|nppc|particles" log | head -20; du -sh . | tail -1
cp log /root/repo/gpurun_out/trecon_slab_log.txt; cd /; rm -rf $W
