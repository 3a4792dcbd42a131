#!/bin/bash
# SQ counters of the push instances of round 4 (one pass per counter set; tools/pmc_sets.sh) -> profiles/r04_sq_*.txt
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04sq; mkdir -p $O
S="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;GRBM_GUI_ACTIVE"
KERNEL="2, false, false>" timeout -k 10 300 bash tools/pmc_sets.sh r04sq_c2 "$S" --steps 10 --warmup 3 > $O/r04_sq_config2_exact.txt 2>&1; tail -19 $O/r04_sq_config2_exact.txt
KERNEL="false, 3, false, false>" timeout -k 10 300 bash tools/pmc_sets.sh r04sq_c3h "$S" --deck trecon --sort-interval -20 --steps 20 --warmup 10 > $O/r04_sq_config3_slab_charged_by_tile_only.txt 2>&1; tail -19 $O/r04_sq_config3_slab_charged_by_tile_only.txt
KERNEL="<true, false, 0, false, false>" timeout -k 10 300 bash tools/pmc_sets.sh r04sq_c3c "$S" --deck trecon --sort-interval -20 --steps 20 --warmup 10 > $O/r04_sq_config3_slab_charge0_copies.txt 2>&1; tail -19 $O/r04_sq_config3_slab_charge0_copies.txt
