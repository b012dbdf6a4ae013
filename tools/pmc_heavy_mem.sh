#!/bin/bash
# Memory-side counters of the splat kernel on the accumulated-cloud frame of tools/tile_hist.py: HBM read bytes, L2 hits
# and misses, requests from the L1s, address-unit and L1 stalls.  One pass per set.  bash tools/pmc_heavy_mem.sh [tag] [sets]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}
SETS=${2:-"0 1 2 3 4"}
PASS[0]="FETCH_SIZE"
PASS[1]="TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"
PASS[2]="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
PASS[3]="TA_TA_BUSY_sum TA_FLAT_LOAD_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE"
PASS[4]="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"
for i in $SETS; do
  OUT=gpurun_out/pmc_heavy_mem_${TAG}_$i
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --pmc ${PASS[$i]} --output-format csv -d $OUT -- python3 tools/tile_hist.py 64 > $OUT/log.txt 2>&1
  python3 tools/pmc_summary.py $OUT | grep -E "splat|fill|project_count" | cut -c1-400
done
