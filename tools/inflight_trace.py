"""Reads a rocprofv3 kernel trace of the bench (tools/inflight_trace.sh) and reports, for the timed part of the run, each
kernel's duration with frames in flight, how much of the wall time some launch of it was running, and the mean number of
this library's kernels running at once."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

src = Path(sys.argv[1])
f = max(src.rglob("*kernel_trace.csv"), key=lambda p: p.stat().st_mtime)
rows = []
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        if "ep::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ep::", "").split("<")[0].strip()
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
splats = [r for r in rows if r[2] == "k_splat_sweep"]
# the last 200 splat launches are the event pass on one stream (bench.py's per-kernel timing); the 60 before them are timed
n_tail = int(sys.argv[2]) if len(sys.argv) > 2 else 60
timed = splats[-(n_tail + 60):-n_tail] if len(splats) >= n_tail + 60 else splats
t0, t1 = timed[0][0], timed[-1][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
wall = t1 - t0
print(f"{len(timed)} frames in {wall / 1e3:.0f} us: {wall / 1e3 / len(timed):.1f} us per frame; queues: {sorted(set(r[3] for r in sel))}")
by = defaultdict(list)
for s, e, n, q in sel:
    by[n].append((s, e))
for n, iv in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    dur = [e - s for s, e in iv]
    iv.sort()
    cover, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += cur_e - cur_s
    print(f"  {n:24s} launches {len(iv):4d}  mean {sum(dur) / len(dur) / 1e3:7.1f} us  max {max(dur) / 1e3:7.1f}  "
          f"sum/wall {sum(dur) / wall:5.2f}  covered {cover / wall:5.2f}")
ev = sorted([(s, 1) for s, e, n, q in sel] + [(e, -1) for s, e, n, q in sel])
area, level, last = 0, 0, t0
for t, d in ev:
    area += level * (t - last)
    level += d
    last = t
print(f"mean kernels running at once: {area / wall:.2f}")
