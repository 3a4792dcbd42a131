#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
KERNEL="2, false, true>" bash tools/pmc_sets.sh r04g_sortpush_tcc "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE" --steps 12 --warmup 9
