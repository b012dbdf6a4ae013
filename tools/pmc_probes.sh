#!/bin/bash
# Counters of the splat kernel's cumulative-phase probes: bash tools/pmc_probes.sh "<counters>" [bench args]
# (probe builds: 10 load, 14 + level-1 histogram, 15 + scatter, 11 + ranking = sorted, 12 + sweep, 13 + merge, 0 everything)
set -e
CTR=${1:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"}
shift || true
for v in 10 14 15 11 12 13 0; do
  echo "variant $v: $(bash tools/pmc_pass.sh p$v "$CTR" --opt splat_variant=$v "$@" | grep splat | sed 's/.*n= *[0-9]* //')"
done
