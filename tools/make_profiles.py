"""Turns the raw output of tools/profile_round.sh (gpurun_out/prof_<round>/) into the committed summaries under profiles/:
  <round>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the profiled command, every kernel of the process
  <round>_kernel_stats_ep.csv   ... this library's kernels only
  <round>_pmc_sq.txt            SQ / GRBM counters, per-launch means
  <round>_pmc_lds.txt           LDS counters, per-launch means
  <round>_pmc_hbm.txt           FETCH_SIZE / WRITE_SIZE (KiB), per-launch means
  pmc_traffic.json              all of the above per kernel, the command they were collected on and the frames it renders;
                                bench.py reads the dominant kernel's entry for its `roofline` object
Per-launch means leave out each kernel's first `warmup` launches (the command's untimed warm-up frames), so that the
counters cover exactly the frames the timing covers.
usage: python tools/make_profiles.py r03"""
import csv
import json
import re
import shutil
import sys
from collections import defaultdict
from pathlib import Path

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = Path(__file__).resolve().parents[1]
src = ROOT / "gpurun_out" / f"prof_{R}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)
command = (src / "command.txt").read_text().strip() if (src / "command.txt").exists() else "bench.py"
m_steps, m_warm = re.search(r"--steps (\d+)", command), re.search(r"--warmup (\d+)", command)
steps, warmup = int(m_steps.group(1)) if m_steps else 200, int(m_warm.group(1)) if m_warm else 20


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
    """{kernel: {counter: mean over the launches after the warm-up}}, and the number of launches averaged."""
    per = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [(dispatch id, value)]
    f = newest((src / sub).rglob("*counter_collection.csv"))
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if "ep::" in row["Kernel_Name"]:
                per[short(row["Kernel_Name"])][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    out, n = {}, {}
    for k, d in per.items():
        out[k] = {}
        for c, v in d.items():
            v = [x for _, x in sorted(v)]
            v = v[warmup:] if len(v) > warmup else v
            out[k][c] = sum(v) / len(v)
            n[k] = len(v)
    return out, n


fetch, n_f = counters("pmc_fetch")
write, _ = counters("pmc_write")
sq, n_sq = counters("pmc_sq") if (src / "pmc_sq").exists() else ({}, {})
lds, _ = counters("pmc_lds") if (src / "pmc_lds").exists() else ({}, {})
out = {"_note": "Per-launch means of rocprofv3 --pmc passes (one pass per counter group, no trace alongside) over the launches "
                "of the timed frames.  HBM bytes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte "
                "read request as 64 bytes, so streaming reads (16 bytes per lane) are 2 x FETCH_SIZE; for scattered 8- and "
                "16-byte gathers the factor lies between 1 and 2, hence both figures are kept: bytes_corrected = "
                "(2 FETCH + WRITE) * 1024, bytes_uncorrected = (FETCH + WRITE) * 1024.  SQ cycle counters are in units of "
                "4 clocks.",
       "_round": R, "_command": "python3 " + command, "_steps": steps, "_warmup": warmup,
       "_frames": f"exo / ego frames {warmup}..{warmup + steps - 1} of the synthetic clip's orbit (the first {warmup} launches of "
                  "every kernel, the warm-up frames, are left out of the means)",
       "kernels": {}}
for k in sorted(set(fetch) | set(write) | set(sq)):
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
    w_ = write.get(k, {}).get("WRITE_SIZE", 0.0)
    out["kernels"][k] = {"launches_averaged": n_f.get(k, n_sq.get(k, 0)),
                         "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w_, 1),
                         "bytes_corrected": int((2 * f + w_) * 1024), "bytes_uncorrected": int((f + w_) * 1024),
                         "sq": {c: int(v) for c, v in sorted(sq.get(k, {}).items())},
                         "lds": {c: int(v) for c, v in sorted(lds.get(k, {}).items())}}
with open(dst / f"{R}_pmc_hbm.txt", "w") as fh:
    fh.write(f"# {out['_command']}; {out['_frames']}\n")
    for k, d in out["kernels"].items():
        fh.write(f"{k:28s} FETCH_SIZE={d['FETCH_SIZE_KiB']} KiB  WRITE_SIZE={d['WRITE_SIZE_KiB']} KiB  launches={d['launches_averaged']}\n")
for name, table in (("sq", sq), ("lds", lds)):
    if table:
        with open(dst / f"{R}_pmc_{name}.txt", "w") as fh:
            fh.write(f"# {out['_command']}; {out['_frames']}\n")
            for k, d in sorted(table.items()):
                fh.write(f"{k:28s} " + "  ".join(f"{c}={v:.6g}" for c, v in sorted(d.items())) + "\n")
if (src / "bench_plain.json").exists():
    shutil.copy(src / "bench_plain.json", dst / f"{R}_bench_same_command.json")
json.dump(out, open(dst / "pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: (v["bytes_corrected"], v["sq"].get("SQ_INSTS_VALU")) for k, v in out["kernels"].items()}, indent=1))
