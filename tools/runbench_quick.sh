for cfg in "$@"; do
  name=$(echo "$cfg" | tr ' =' '__' | tr -d '-')
  timeout -k 10 300 python -u bench.py --steps 60 --warmup 10 --frames 64 --skip-cpu $cfg > gpurun_out/b_$name.json 2> gpurun_out/b_$name.err || { tail -5 gpurun_out/b_$name.err; exit 1; }
  python - "$cfg" gpurun_out/b_$name.json <<PY
import json,sys
d=json.load(open(sys.argv[2]))
print(sys.argv[1], "|", d["value"], "fps", {k.replace("k_",""):v["avg_us"] for k,v in d["kernels"].items() if "splat" in k})
PY
done
