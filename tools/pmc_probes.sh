#!/bin/bash
# Instruction counts of the splat kernel's cumulative-phase probes: bash tools/pmc_probes.sh [bench args]
set -e
for v in 10 14 11 15 12 13 0; do
  echo "variant $v: $(bash tools/pmc_pass.sh p$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" --opt splat_variant=$v "$@" | grep splat | sed 's/.*n= *[0-9]* //')"
done
