set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ' =' '__' | tr -d '-')
  timeout -k 10 300 python -u bench.py --steps 100 --warmup 10 --skip-cpu $cfg > gpurun_out/b_$name.json 2> gpurun_out/b_$name.err || { tail -5 gpurun_out/b_$name.err; exit 1; }
  python - "$cfg" gpurun_out/b_$name.json <<PY
import json,sys
d=json.load(open(sys.argv[2]))
print(sys.argv[1], "|", d["value"], "fps", d["gpoints_per_s"], "Gpts/s", {k.replace("k_",""):v["avg_us"] for k,v in d["kernels"].items()})
PY
done
