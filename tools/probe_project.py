"""Diagnostic: time of the projection kernel alone (no tile counting) on the bench's static cloud."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "egox-egopriorrenderer_amd")]
import numpy as np
import torch
from egoprior import _lib, engine as eng, synth

E = eng.Engine("cuda:0")
H, W, He, We = 1080, 1920, 480, 832
d, rgb, m = synth.make_frames(H, W, [12], 300, device="cuda:0")
exo = synth.exo_w2c()
T = np.linalg.inv(exo)
intr, c2w = synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32)
bg = E.unproject_frame(d[0], rgb[0], None, intr, c2w, T, subsample=1, select=_lib.SELECT_ALL)
R, t = eng.split_w2c(synth.ego_orbit_w2c(300, exo)[12])
cam = eng.pinhole_camera(R, t, synth.ego_pinhole_K(He, We), He, We)
n = bg.n()
for rep in range(3):
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(20):
        E.project_points(cam, bg)
    k = _lib.profile_end()
    for name, (ms, calls) in k.items():
        us = 1e3 * ms / calls
        print(f"{name}: {us:.2f} us per launch, {n} points, {32 * n / us / 1e3:.1f} GB/s ({32 * n / us / 8e6 * 100:.1f}% of 8 TB/s)")
