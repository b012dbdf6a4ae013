"""Diagnostic: time of the projection + tile-counting kernel alone (egoprior_count_list_entries on the bench's static
cloud; no records are written, so a library built with -DEP_BIN_PROBE=n, which leaves parts of the counting out, is safe
to run here and nowhere else).  EGOPRIOR_LIB picks the library."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "egox-egopriorrenderer_amd")]
import numpy as np
import torch
from egoprior import _lib, engine as eng, synth

E = eng.Engine("cuda:0")
L = _lib.lib()
H, W, He, We = 1080, 1920, 480, 832
d, rgb, m = synth.make_frames(H, W, [12], 300, device="cuda:0")
exo = synth.exo_w2c()
T = np.linalg.inv(exo)
intr, c2w = synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32)
bg = E.unproject_frame(d[0], rgb[0], None, intr, c2w, T, subsample=1, select=_lib.SELECT_ALL)
R, t = eng.split_w2c(synth.ego_orbit_w2c(300, exo)[12])
cam = eng.pinhole_camera(R, t, synth.ego_pinhole_K(He, We), He, We)
rp = eng.make_raster_params(5.0, He, We)
n = bg.capacity
ws = torch.empty(L.egoprior_raster_workspace_bytes(He, We, n, rp.radius), dtype=torch.uint8, device="cuda:0")
entries = torch.zeros(1, dtype=torch.int64, device="cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(2):
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(20):
        _lib.check(L.egoprior_count_list_entries(C.byref(cam), C.byref(rp), He, We, C.c_void_p(bg.points.data_ptr()), n,
                                                 C.c_void_p(bg.count.data_ptr()), None, 0, None,
                                                 C.c_void_p(entries.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), st))
    k = _lib.profile_end()
print(f"{bg.n()} points, {int(entries.item())} list entries; " +
      ", ".join(f"{nm} {1e3 * ms / calls:.2f} us" for nm, (ms, calls) in k.items()))
