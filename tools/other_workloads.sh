#!/bin/bash
# The other workloads quoted next to the headline number (60-step runs, 4 frames in flight).  Run on the GPU box:
# bash tools/other_workloads.sh <tag>
set -e
T=${1:-r02}
O=gpurun_out/other_$T
mkdir -p $O
B="timeout -k 10 200 python bench.py --steps 60 --warmup 10 --skip-cpu --frames 64"
$B --point-size 1 > $O/ps1.json 2> $O/ps1.err
$B --point-size 10 > $O/ps10.json 2> $O/ps10.err
$B --point-size 20 > $O/ps20.json 2> $O/ps20.err
$B --fisheye --ego 448x448 > $O/fe5.json 2> $O/fe5.err
$B --fisheye --ego 448x448 --point-size 10 > $O/fe10.json 2> $O/fe10.err
$B --fisheye --ego 448x448 --point-size 20 > $O/fe20.json 2> $O/fe20.err
$B --subsample 2 > $O/sub2.json 2> $O/sub2.err
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --skip-cpu --frames 16 --exo 2160x3840 --ego 1024x1024 > $O/k4.json 2> $O/k4.err
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --skip-cpu --frames 64 --streams 1 > $O/serial.json 2> $O/serial.err
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --skip-cpu --frames 64 --from-host --depth-f16 > $O/fromhost16.json 2> $O/fromhost16.err
timeout -k 10 300 python -u tools/cli_timing.py 64 --fisheye > $O/cli_fisheye.log 2>&1
echo done
