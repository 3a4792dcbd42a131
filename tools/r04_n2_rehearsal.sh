#!/bin/bash
# bench.py's N > 1 plumbing on the one GPU of a box: two ranks share the device, gloo moves the messages (host-staged) --
# RCCL refuses two ranks on one device; the RCCL transport itself is tested with self-sends (tests/test_gpu_rccl_host.py)
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1 VPIC_HIP_SINGLE_DEVICE=1
O=gpurun_out/r04n2; mkdir -p $O
run() { timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --backend gloo --no-cpu-baseline "${@:2}" 2> $O/err_$1.txt | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('  %.2f G pushes/s  %.2f ms/step  transport %s  check %s' % (j['value']/1e9, j['ms_per_step'], j.get('transport'), {k: v for k, v in (j.get('check') or {}).items() if k in ('particles_conserved',)}))
print('   exchange', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (j.get('exchange') or {}).items() if k != 'max_over_ranks'})
for s in j.get('advance_p_by_species') or []: print('     species %d charged %s: %.3f ms/launch' % (s['species'], s['charged'], s['avg_launch_ms']))"; tail -3 $O/err_$1.txt | cut -c1-300; }
echo "-- two-stream 128^3 x 32 ppc in two x-slabs"; run 29511 --config 1 --steps 10 --warmup 3
echo "-- configs[3] in small (64 x 64 x 32, walls in z, 4 species) in two x-slabs"; run 29512 --deck trecon --grid 64 64 32 --ppc 32 --sort-interval -20 --steps 10 --warmup 3
echo "-- 2 x 1 x ... bricks: 64^3 in 1 x 2 x 1"; run 29513 --grid 64 64 64 --ppc 32 --topology 1 2 1 --steps 10 --warmup 3
echo "-- configs[4] in small (cold uniform drift, 1 species x 128 ppc, 64^3) in two x-slabs"; run 29514 --deck drift --grid 64 64 64 --ppc 128 --steps 10 --warmup 3
