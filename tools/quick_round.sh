#!/bin/bash
# One GPU round for a kernel change: raster parity tests, heavy accumulated frame, two short bench runs, CLI timing.
# bash tools/quick_round.sh <tag>
T=${1:-x}
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/q_${T}_tests.log 2>&1 || { tail -15 gpurun_out/q_${T}_tests.log; exit 1; }
tail -1 gpurun_out/q_${T}_tests.log
timeout -k 10 120 python3 tools/tile_hist.py 64 > gpurun_out/q_${T}_heavy.log 2>&1 || exit 1
tail -1 gpurun_out/q_${T}_heavy.log
for i in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 100 --skip-cpu > gpurun_out/q_${T}_bench$i.json 2> gpurun_out/q_${T}_bench$i.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/q_${T}_bench$i.json'))
print('bench', d['value'], 'fps; splat', d['kernels'].get('k_splat_sweep',{}).get('avg_us'), 'serial-kernel-sum', round(sum(k['avg_us'] for k in d['kernels'].values()),1))"
done
timeout -k 10 200 python3 tools/cli_timing.py > gpurun_out/q_${T}_cli.log 2>&1 || exit 1
grep -i "best of\|frames/s" gpurun_out/q_${T}_cli.log | tail -2
