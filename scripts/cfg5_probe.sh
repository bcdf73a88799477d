#!/bin/bash
# cfg5-shaped solve (1920x1080 -> 3840x2160, 17 frames): kernel times with and without camera rotation, then counter passes
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
A="--rows 1080 --cols 1920 --scale 2 --K 17 --iters 6 --reps 1"
timeout -k 10 100 python3 scripts/quick_time.py $A | tail -1 || exit 1
timeout -k 10 100 python3 scripts/quick_time.py $A --rot 0 | tail -1 || exit 1
timeout -k 10 100 python3 scripts/quick_time.py $A --rot 0 --shift 0 | tail -1 || exit 1
export QT_ARGS="--rows 1080 --cols 1920 --scale 2 --K 17 --iters 3 --reps 1"
scripts/pmc2.sh c5sq default SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU || exit 1
scripts/pmc2.sh c5tcc default TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || exit 1
scripts/pmc2.sh c5tcp default TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum
