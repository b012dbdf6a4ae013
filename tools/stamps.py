"""Diagnostic: per-phase cycle shares of k_splat_sweep (needs csrc/libegoprior_hip_stamps.so, built with
`make -C egox-egopriorrenderer_amd/csrc libegoprior_hip_stamps.so`).  Not part of tests or bench."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "egox-egopriorrenderer_amd")]
import numpy as np
import torch
from egoprior import _lib

_lib.LIB_PATH = _lib.CSRC / "libegoprior_hip_stamps.so"
from egoprior import engine as eng, synth

ps = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
E = eng.Engine("cuda:0")
_lib.set_option("splat_variant", variant)
dbg = C.CDLL(str(_lib.LIB_PATH)).egoprior_debug_stamps
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
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    E.render_frame(cam, rp, bg)
ev1.record()
torch.cuda.synchronize()
print(f"render_frame: {ev0.elapsed_time(ev1) / 10 * 1e3:.1f} us per frame (diagnostic build)")
buf = (C.c_ulonglong * 16)()
dbg(buf, 1)
E.render_frame(cam, rp, bg)
torch.cuda.synchronize()
dbg(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = ["loop-top", "load", "bucket+scatter", "sort", "mask-build", "append", "sweep-exit", "merge", "exit", "composite"]
tot = v[:8].sum()
for n, x in zip(names, v[:8]):
    print(f"{n:16s} {x:14.0f} cycles  {100 * x / tot:5.1f}%")
print(f"bitonic fallbacks {v[13]:.0f} of {v[12]:.0f} chunks; mean longest per-lane run {v[14] / max(v[12], 1):.1f}")
print(f"chunks by largest bucket: >4 {v[8]:.0f}, >8 {v[9]:.0f}, >32 {v[15]:.0f}, >256 (bitonic) {v[13]:.0f}")
print(f"blocks swept {v[10]:.0f} of {v[11]:.0f} ({100 * v[10] / max(v[11], 1):.1f}%), chunks {v[12]:.0f}, total wave-cycles {tot:.3g}")
