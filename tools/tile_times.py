"""Diagnostic: duration of every k_splat_sweep wave (needs csrc/libegoprior_hip_tiletimes.so, built with
`make -C egox-egopriorrenderer_amd/csrc libegoprior_hip_tiletimes.so`).  Not part of tests or bench."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "egox-egopriorrenderer_amd")]
import numpy as np
import torch
from egoprior import _lib

import os
_lib.LIB_PATH = _lib.CSRC / os.environ.get("EGOPRIOR_DIAG_LIB", "libegoprior_hip_tiletimes.so")
from egoprior import engine as eng, synth

ps = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
E = eng.Engine("cuda:0")
_lib.set_option("splat_variant", variant)
dbg = C.CDLL(str(_lib.LIB_PATH)).egoprior_debug_tile_times
H, W, He, We = 1080, 1920, 480, 832
d, rgb, m = synth.make_frames(H, W, [12], 300, device="cuda:0")
exo = synth.exo_w2c()
T = np.linalg.inv(exo)
intr, c2w = synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32)
bg = E.unproject_frame(d[0], rgb[0], None, intr, c2w, T, subsample=1, select=_lib.SELECT_STATIC)
pose = int(sys.argv[3]) if len(sys.argv) > 3 else 12
R, t = eng.split_w2c(synth.ego_orbit_w2c(300, exo)[pose])
cam = eng.pinhole_camera(R, t, synth.ego_pinhole_K(He, We), He, We)
rp = eng.make_raster_params(ps, He, We, rotate_cw=True)
E.render_frame(cam, rp, bg)
torch.cuda.synchronize()
cs = C.CDLL(str(_lib.LIB_PATH)).egoprior_debug_stage_cycles
cbuf = (C.c_uint64 * 32)()
cs(cbuf, 1)
E.render_frame(cam, rp, bg)
torch.cuda.synchronize()
cs(cbuf, 0)
c = [int(v) for v in cbuf]
tot = max(sum(c[:5]), 1)
print("wave cycles by stage (all waves): " + ", ".join(
    f"{n} {100 * v / tot:.1f}%" for n, v in zip(("load+filter", "sort", "sweep+append", "merge", "compositor"), c[:5]))
    + f"; total {tot:.4g} over {c[5]} waves; absolute (Mcycles): " + ", ".join(f"{v / 1e6:.1f}" for v in c[:5]))
print(f"chunks {c[6]}, of which bitonic {c[7]}; buckets ranked by the wave {c[8]} holding {c[9]} records; records sorted {c[10]}")
print(f"gather: {c[15]} batches; cycles waiting for keys {c[11]:.4g}, for x, y {c[12]:.4g}, tests {c[13]:.4g}")
print(f"merge stage: mask walk {c[16]:.4g}, key reads + network {c[17]:.4g}, tile threshold {c[18]:.4g}, band tables {c[19]:.4g}")
print(f"sweep: {c[25]} blocks offered; lane-per-record form {c[20]} blocks ({c[23] / max(c[20], 1):.1f} open pixels each), "
      f"open-pixel form {c[21]} blocks ({c[22] / max(c[21], 1):.1f} open pixels each), early exits {c[24]}")
n_tiles = ((He + 7) // 8) * ((We + 7) // 8)
buf = (C.c_uint32 * (4 * n_tiles))()
assert dbg(buf, 4 * n_tiles) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(-1, 4).astype(np.int64)
tile, rec, cyc, chunks = a.T
print(f"{n_tiles} waves: records mean {rec.mean():.0f} max {rec.max()}; cycles mean {cyc.mean():.0f} max {cyc.max()} "
      f"sum {cyc.sum():.3g}; chunks max {chunks.max()}")
order = np.argsort(-cyc)[:12]
for o in order:
    print(f"  launch slot {o:5d} tile {tile[o]:5d} records {rec[o]:6d} chunks {chunks[o]:2d} cycles {cyc[o]:8d}")
qs = np.percentile(cyc, [50, 90, 99, 99.9])
print("cycle percentiles 50/90/99/99.9:", [int(q) for q in qs])
bins = [0, 256, 512, 1024, 2048, 4096, 8192, 1 << 30]
for lo, hi in zip(bins[:-1], bins[1:]):
    sel = (rec >= lo) & (rec < hi)
    if sel.any():
        print(f"  records [{lo},{hi}): {sel.sum():5d} tiles, mean cycles {cyc[sel].mean():9.0f}, cycles/record {cyc[sel].sum() / max(rec[sel].sum(), 1):6.1f}, share of all cycles {100 * cyc[sel].sum() / cyc.sum():5.1f}%")
