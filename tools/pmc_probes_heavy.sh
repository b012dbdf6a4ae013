#!/bin/bash
# Like pmc_probes.sh, on the accumulated-cloud frame of tools/tile_hist.py (31.5 M points, 448x448, point size 5).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 11 12 13 14 0; do
  OUT=gpurun_out/pmc_heavy_$v
  rm -rf $OUT; mkdir -p $OUT
  EGOPRIOR_LIB=libegoprior_hip_tiletimes.so EGOPRIOR_OPTIONS=splat_variant=$v timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 tools/tile_hist.py 64 > $OUT/log.txt 2>&1
  echo "variant $v: $(python3 tools/pmc_summary.py $OUT | grep splat | sed 's/.*n= *[0-9]* //')"
done
