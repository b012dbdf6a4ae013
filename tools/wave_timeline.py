"""Diagnostic: when every wave of one k_splat_sweep launch started and ended (needs csrc/libegoprior_hip_spans.so:
`make -C egox-egopriorrenderer_amd/csrc libegoprior_hip_spans.so`, the product kernels plus two counter reads per wave).
usage: python tools/wave_timeline.py light|heavy [point_size]"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "egox-egopriorrenderer_amd")]
import numpy as np
import torch
from egoprior import _lib

_lib.LIB_PATH = _lib.CSRC / os.environ.get("EGOPRIOR_DIAG_LIB", "libegoprior_hip_spans.so")
from egoprior import engine as eng, synth

which = sys.argv[1] if len(sys.argv) > 1 else "light"
ps = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
E = eng.Engine("cuda:0")
exo = synth.exo_w2c()
T = np.linalg.inv(exo)
H, W = 1080, 1920
intr, c2w = synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32)
if which == "light":   # one 1080p frame at subsample 1 -> 480x832 (the bench workload)
    He, We = 480, 832
    d, rgb, m = synth.make_frames(H, W, [12], 300, device="cuda:0")
    bg = E.unproject_frame(d[0], rgb[0], None, intr, c2w, T, subsample=1, select=_lib.SELECT_STATIC)
    R, t = eng.split_w2c(synth.ego_orbit_w2c(300, exo)[12])
    cam = eng.pinhole_camera(R, t, synth.ego_pinhole_K(He, We), He, We)
else:                  # static cloud accumulated over 64 frames at subsample 2 -> 448x448 (the command-line tool's default)
    He, We, F = 448, 448, 64
    d, rgb, m = synth.make_frames(H, W, range(F), F, device="cuda:0")
    clouds = [E.unproject_frame(d[f], rgb[f], m[f], intr, c2w, T, subsample=2, select=_lib.SELECT_STATIC) for f in range(F)]
    pts = torch.cat([c.points[:c.n()] for c in clouds])
    bg = E.new_cloud(pts.shape[0])
    bg.points.copy_(pts)
    bg.count.fill_(pts.shape[0])
    bg.flags.fill_(1)
    K = np.array([[150.0, 0, 224.0], [0, 150.0, 224.0], [0, 0, 1]], dtype=np.float32)
    R, t = eng.split_w2c(synth.ego_orbit_w2c(F, exo)[F // 2])
    cam = eng.pinhole_camera(R, t, K, He, We)
rp = eng.make_raster_params(ps, He, We, rotate_cw=True)
spans = C.CDLL(str(_lib.LIB_PATH)).egoprior_debug_wave_spans
buf = (C.c_uint32 * (4 * 16384))()
for _ in range(2):
    E.render_frame(cam, rp, bg)
torch.cuda.synchronize()
assert spans(buf, 16384, 1) == 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
E.render_frame(cam, rp, bg)
ev1.record()
torch.cuda.synchronize()
assert spans(buf, 16384, 0) == 0
cbuf = (C.c_uint32 * (4 * 16384))()
assert C.CDLL(str(_lib.LIB_PATH)).egoprior_debug_wave_chunk_ends(cbuf, 16384) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(-1, 4).astype(np.int64)
ce = np.frombuffer(cbuf, dtype=np.uint32).reshape(-1, 4).astype(np.int64)
ce = ce[a[:, 1] != 0]
a = a[a[:, 1] != 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) & 0xFFFFFFFF, (a[:, 1] - t0) & 0xFFFFFFFF
rec, chunks = a[:, 2], a[:, 3]
span, dur = en.max(), en - st
print(f"{which}: frame {ev0.elapsed_time(ev1) * 1000:.0f} us; {len(a)} waves; splat launch {span / 100.0:.1f} us; "
      f"wave time = {100 * dur.sum() / (span * 3072.0):.1f}% of 3072 slots; sum / slots = {dur.sum() / 3072.0 / 100.0:.1f} us")
edges = np.linspace(0, span, 21)
print("waves alive per 5% of the launch:", [int(((st < b) & (en > a_)).sum()) for a_, b in zip(edges[:-1], edges[1:])])
print("wave duration percentiles 50/90/99/max (us):", [round(float(np.percentile(dur, q)) / 100.0, 1) for q in (50, 90, 99, 100)])
for name, order in (("longest", np.argsort(-dur)[:6]), ("last to finish", np.argsort(-en)[:6])):
    print(f"{name} (duration us, start us, entries, chunks):",
          [(round(float(dur[o]) / 100.0, 1), round(float(st[o]) / 100.0, 1), int(rec[o]), int(chunks[o])) for o in order])
for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 6), (6, 1 << 30)):
    sel = (chunks >= lo) & (chunks < hi)
    if sel.any():
        print(f"  chunks [{lo},{hi}): {int(sel.sum()):5d} waves, mean {dur[sel].mean() / 100.0:6.1f} us, share of wave time {100 * dur[sel].sum() / dur.sum():5.1f}%")
for nc in (2, 3, 4):
    sel = chunks == nc
    if sel.sum() < 10:
        continue
    ends = (ce[sel, :nc] - t0) & 0xFFFFFFFF
    prev = np.concatenate([st[sel, None], ends[:, :-1]], axis=1)
    per = (ends - prev).mean(axis=0) / 100.0
    rest = (en[sel] - ends[:, -1]).mean() / 100.0
    print(f"  waves with {nc} chunks: mean us per chunk {[round(float(v), 1) for v in per]}, after the last chunk (merges, compositor) {rest:.1f}")
