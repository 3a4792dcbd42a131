#!/bin/bash
# second GPU pass of round 4: the whole GPU suite (RCCL self-send tests of the C++ deck host included), the one-launch
# ablation ledger of advance_p (tools/ablate_once.py) cold and hot, configs[1] as a deck on the C++ host
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
echo "-- one-launch ablation, 256^3 x 64 ppc"
VPIC_HIP_LIB=$PWD/tools/ab/libablation.so timeout -k 10 400 python tools/ablate_once.py 0 256 512 32 288 64 2 > $O/ablate_once_cold.txt 2>&1; cat $O/ablate_once_cold.txt
echo "-- one-launch ablation, configs[3] slab"
VPIC_HIP_LIB=$PWD/tools/ab/libablation.so timeout -k 10 300 python tools/ablate_once.py --deck trecon --steps-before 8 0 256 512 32 288 64 2 > $O/ablate_once_hot.txt 2>&1; cat $O/ablate_once_hot.txt
echo "-- configs[1] as a deck on the C++ host (adaptive sorting = the host's default, then fixed intervals)"
mkdir -p $O/deck && cd $O/deck
VPIC_HIP_HOST_TIMING=1 timeout -k 10 300 ../../../old-vpic_amd/host/twostream128.hip.exe -tpp=1 40 2>&1 | grep -i "simulation time\|hip host" 
VPIC_HIP_ADAPTIVE_SORT=0 VPIC_HIP_HOST_TIMING=1 timeout -k 10 300 ../../../old-vpic_amd/host/twostream128.hip.exe -tpp=1 40 2>&1 | grep -i "simulation time\|hip host"
