"""Turns the raw output of tools/profile_round.sh (gpurun_out/prof_<round>/) into the committed summaries under profiles/:
<round>_kernel_stats.csv (all kernels), <round>_kernel_stats_ep.csv (this library's kernels), pmc_traffic.json (HBM
bytes per launch, gfx950 read-side correction applied) and <round>_pmc_sq.txt (SQ instruction counters per launch).
usage: python tools/make_profiles.py r01"""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = Path(__file__).resolve().parents[1]
src = ROOT / "gpurun_out" / f"prof_{R}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ep::", "").split("<")[0].strip()


def newest(paths):
    """gpurun merges every run's files into the same local directory: keep the most recent one."""
    return max(paths, key=lambda f: f.stat().st_mtime)


stats = newest((src / "trace").rglob("*kernel_stats.csv"))
shutil.copy(stats, dst / f"{R}_kernel_stats.csv")
with open(stats, newline="") as fh:
    rows = list(csv.reader(fh))
with open(dst / f"{R}_kernel_stats_ep.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(rows[0])
    for r in rows[1:]:
        if "ep::" in r[0]:
            w.writerow(r)


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in [newest((src / sub).rglob("*counter_collection.csv"))]:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "ep::" in row["Kernel_Name"]:
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


fetch, write = counters("pmc_fetch"), counters("pmc_write")
out = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB units), bench.py "
                "workload (1080p exo, 480x832 ego, point_size 5), one frame in flight. gfx950 correction applied: "
                "FETCH_SIZE counts 128-B read requests as 64 B, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is "
                "exact for 16-B-per-lane stores.",
       "_round": R}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
    w_ = write.get(k, {}).get("WRITE_SIZE", 0.0)
    out[k] = int(2 * f * 1024 + w_ * 1024)
    out[k + "__raw_KiB"] = {"FETCH_SIZE": round(f, 1), "WRITE_SIZE": round(w_, 1)}
sq = counters("pmc_sq") if (src / "pmc_sq").exists() else {}
if sq:
    out["_sq_per_launch"] = {k: {c: int(v) for c, v in d.items()} for k, d in sorted(sq.items())}
    with open(dst / f"{R}_pmc_sq.txt", "w") as fh:
        for k, d in sorted(sq.items()):
            fh.write(f"{k:28s} " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())) + "\n")
lds = counters("pmc_lds") if (src / "pmc_lds").exists() else {}
if lds:
    with open(dst / f"{R}_pmc_lds.txt", "w") as fh:
        for k, d in sorted(lds.items()):
            fh.write(f"{k:28s} " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())) + "\n")
json.dump(out, open(dst / "pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_") and not k.endswith("KiB")}, indent=1))
