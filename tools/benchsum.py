#!/usr/bin/env python3
"""Prints the headline figures of one or more bench.py JSON lines: frames/s, ms per step and per-kernel averages."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().split("\n")[-1])
    except Exception as e:  # a run that crashed leaves no JSON
        print(f"{path}: unreadable ({e})")
        continue
    ks = " ".join(f"{k.replace('k_', '')}={v['avg_us']:.1f}" for k, v in d.get("kernels", {}).items())
    print(f"{path}: {d['value']:.0f} fps  {1e3 * d['ms_per_step']:.1f} us/step  streams={d['config'].get('frames_in_flight')}  | {ks}")
