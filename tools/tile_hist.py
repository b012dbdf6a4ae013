"""Diagnostic: how the points of the command-line tool's default workload (static cloud accumulated over F 1080p frames,
448x448 ego frames) spread over the 8x8 tiles, by disc centre.  usage: python tools/tile_hist.py [frames] [pose]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "egox-egopriorrenderer_amd")]
import numpy as np
import torch

from egoprior import _lib, engine as eng, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pose = int(sys.argv[2]) if len(sys.argv) > 2 else F // 2
E = eng.Engine("cuda:0")
H, W, He, We = 1080, 1920, 448, 448
d, rgb, m = synth.make_frames(H, W, range(F), F, device="cuda:0")
fx = synth.load_camera_fixture()
exo = synth.exo_w2c(fx)
T = np.linalg.inv(exo)
intr, c2w = synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32)
clouds = [E.unproject_frame(d[f], rgb[f], m[f], intr, c2w, T, subsample=2, select=_lib.SELECT_STATIC) for f in range(F)]
pts = torch.cat([c.points[:c.n()] for c in clouds])
bg = E.new_cloud(pts.shape[0])
bg.points.copy_(pts)
bg.count.fill_(pts.shape[0])
bg.flags.fill_(1)
K = np.array([[150.0, 0, 224.0], [0, 150.0, 224.0], [0, 0, 1]], dtype=np.float32)
R, t = eng.split_w2c(synth.ego_orbit_w2c(F, exo)[pose])
cam = eng.pinhole_camera(R, t, K, He, We)
ndc = E.project_points(cam, bg)[: pts.shape[0]]
x, y, z = ndc[:, 0], ndc[:, 1], ndc[:, 2]
# NDC of pytorch3d: +x left, +y up, the shorter side spans [-1, 1]
s = min(He, We)
px = (We - 1) / 2.0 - x * s / 2.0
py = (He - 1) / 2.0 - y * s / 2.0
ok = (z > 0) & (px > -3) & (px < We + 2) & (py > -3) & (py < He + 2)
tx = (px[ok].clamp(0, We - 1) / 8).long()
ty = (py[ok].clamp(0, He - 1) / 8).long()
hist = torch.bincount(ty * (We // 8) + tx, minlength=(He // 8) * (We // 8)).cpu().numpy()
print(f"{pts.shape[0]} points, {int(ok.sum())} in view; tiles {hist.size}, non-empty {(hist > 0).sum()}")
print("records per tile by disc centre: mean %.0f  p50 %d  p90 %d  p99 %d  max %d" % (
    hist.mean(), *np.percentile(hist, [50, 90, 99]).astype(int), hist.max()))
rp = eng.make_raster_params(5.0, He, We, rotate_cw=True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
E.render_frame(cam, rp, bg)
ev0.record()
for _ in range(3):
    E.render_frame(cam, rp, bg)
ev1.record()
torch.cuda.synchronize()
print(f"render_frame: {ev0.elapsed_time(ev1) / 3:.2f} ms")
if "tiletimes" in str(_lib.LIB_PATH):
    import ctypes as C
    D = C.CDLL(str(_lib.LIB_PATH))
    cbuf = (C.c_uint64 * 32)()
    D.egoprior_debug_stage_cycles(cbuf, 1)
    E.render_frame(cam, rp, bg)
    torch.cuda.synchronize()
    D.egoprior_debug_stage_cycles(cbuf, 0)
    c = [int(v) for v in cbuf]
    tot = max(sum(c[:5]), 1)
    print("wave cycles by stage: " + ", ".join(f"{n} {100 * v / tot:.1f}%" for n, v in zip(
        ("gather", "sort", "sweep+append", "merge", "compositor"), c[:5])) + f"; total {tot:.4g} over {c[5]} waves")
    print(f"chunks {c[6]}, of which bitonic {c[7]}; buckets ranked by the wave {c[8]} holding {c[9]} records; records sorted {c[10]}")
    print(f"gather: {c[15]} batches; cycles waiting for keys {c[11]:.4g}, for x, y {c[12]:.4g}, tests {c[13]:.4g}")
    print(f"merge stage: mask walk {c[16]:.4g}, key reads + network {c[17]:.4g}, tile threshold {c[18]:.4g}, band tables {c[19]:.4g}")
    n_tiles = ((He + 7) // 8) * ((We + 7) // 8)
    buf = (C.c_uint32 * (4 * n_tiles))()
    assert D.egoprior_debug_tile_times(buf, 4 * n_tiles) == 0
    a = np.frombuffer(buf, dtype=np.uint32).reshape(-1, 4).astype(np.int64)
    tile, rec, cyc, chunks = a.T
    print(f"list entries: total {rec.sum()} max {rec.max()}; chunks: total {chunks.sum()} max {chunks.max()}")
    for o in np.argsort(-cyc)[:10]:
        print(f"  tile {tile[o]:5d} records {rec[o]:7d} chunks {chunks[o]:4d} cycles {cyc[o]:9d} "
              f"({cyc[o] / max(chunks[o], 1):7.0f} per chunk, {rec[o] / max(chunks[o], 1):6.0f} records per chunk)")
_lib.profile_begin()
for _ in range(3):
    E.render_frame(cam, rp, bg)
torch.cuda.synchronize()
print("per kernel (us): " + ", ".join(f"{nm} {1e3 * ms / calls:.0f}" for nm, (ms, calls) in _lib.profile_end().items()))
