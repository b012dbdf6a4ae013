#!/bin/bash
# A/B of library builds on the bench: bash tools/ab.sh <tag> lib1.so lib2.so ...   (names inside csrc/; two 100-step runs each)
T=$1; shift
for L in "$@"; do
  for i in 1 2; do
    EGOPRIOR_LIB=$L timeout -k 10 200 python3 bench.py --steps 100 --skip-cpu > gpurun_out/ab_${T}_${L}_$i.json 2> gpurun_out/ab_${T}_${L}_$i.err || { echo "$L failed"; tail -3 gpurun_out/ab_${T}_${L}_$i.err; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/ab_${T}_${L}_$i.json'))
print('$L', d['value'], 'fps; splat', d['kernels'].get('k_splat_sweep',{}).get('avg_us'), 'sum', round(sum(k['avg_us'] for k in d['kernels'].values()),1))"
  done
done
