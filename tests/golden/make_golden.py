"""Generates the committed golden fixtures under tests/golden/.  Runs ONLY in the dev container, where the reference
checkout is mounted at /root/reference; the fixtures (data: inputs + expected outputs) are what travels.

What is produced and from where:
  g1_rays_mask.npz ....... outputs of the reference's own vipe/utils/cameras.py (PinholeCameraModel.iproj_disp) and
                           vipe/utils/depth.py (reliable_depth_mask_range), loaded by file path.  depth.py imports
                           vipe.ext.lietorch at module level (needs the compiled CUDA extension); three empty stand-in
                           modules are put in sys.modules for that import only -- nothing from them is called.
  g3_rays_mei_pano.npz ... ray tables of MeiCameraModel / PanoramaCameraModel.iproj_disp (vipe/utils/cameras.py:228-250,
                           366-387) on the pixel grid of scripts/render_vipe_pointcloud.py:384-398.
  g2_nanmean.npz ......... torch.nan_to_num(torch.nanmean(x, dim=0)) on CPU, the two calls the reference makes at
                           scripts/render_vipe_pointcloud.py:466-471.
  g4_pil_luma.npz ........ PIL's Image.convert("L") on random RGB pixels (dark ones over-represented, every grey level
                           around the threshold 30 present): the call data_preprocess/scripts/postprocess_render_dir.py
                           :17-19 makes before thresholding.
  cameras_cmu_bike08_2.json  camera parameters copied from the reference's example data files
                           (data_preprocess/example/annotations/.../1410e2ce-....json: cam01 K + W2C, aria01 K and its
                           first 128 per-frame W2C; takes/cmu_bike08_2/trajectory/online_calibration.jsonl line 0,
                           camera-rgb FisheyeRadTanThinPrism params).

The pytorch3d boundary (projection, rasterize_points, compositor) has NO golden vectors: pytorch3d is not installed and
the reference holds no outputs for it.
"""
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_helpers():
    cameras = _load("ref_cameras", REF / "vipe/utils/cameras.py")
    for n in ("vipe", "vipe.ext", "vipe.ext.lietorch"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["vipe.ext.lietorch"].SE3 = object
    depth = _load("ref_depth", REF / "vipe/utils/depth.py")
    return cameras, depth


def smooth_depth(H, W, rng):
    v, u = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    d = 2.0 + 0.02 * u + 0.03 * v + 0.3 * np.sin(u / 5.0) * np.cos(v / 4.0)
    d = d.astype(np.float16).astype(np.float32)  # EXR HALF round trip
    # a depth step (unreliable band), and special values
    d[:, W // 2:] += np.float32(1.5)
    d[3, 4] = np.nan
    d[10, 20] = np.inf
    d[15, 7] = 0.0
    d[18, 25] = -1.0
    d[0, 0] = 1e-3
    noise = rng.random((H, W), dtype=np.float32)
    d[20:, :8] *= (1.0 + 0.2 * noise[20:, :8])  # rough patch
    return d


def make_g1(cameras, depth_mod):
    rng = np.random.default_rng(0)
    H, W = 24, 32
    out = {}
    for tag, s, intr in (("a", 2, [24.0, 24.0, 16.0, 12.0]), ("b", 1, [23.7, 25.1, 15.3, 11.9]),
                         ("c", 3, [31.0, 29.5, 17.25, 10.5])):
        intr_t = torch.tensor(intr, dtype=torch.float32)
        model = cameras.CameraType.PINHOLE.build_camera_model(intr_t)
        dv, du = torch.meshgrid(torch.arange(H).float()[::s], torch.arange(W).float()[::s], indexing="ij")
        pts, _, _ = model.iproj_disp(torch.ones_like(dv), du, dv)
        rays = pts[..., :3].numpy()
        rays = rays / rays[..., 2:3]
        out[f"rays_{tag}_intr"] = np.asarray(intr, dtype=np.float32)
        out[f"rays_{tag}_s"] = np.int32(s)
        out[f"rays_{tag}"] = rays.astype(np.float32)
    d = smooth_depth(H, W, rng)
    out["depth"] = d
    out["mask"] = depth_mod.reliable_depth_mask_range(torch.from_numpy(d)).numpy()
    # a second, larger smooth case with fp16-rounded plane + random rough blocks
    H2, W2 = 40, 56
    v, u = np.meshgrid(np.arange(H2, dtype=np.float32), np.arange(W2, dtype=np.float32), indexing="ij")
    d2 = (1.0 + 0.05 * u + 0.01 * v).astype(np.float16).astype(np.float32)
    blk = rng.random((H2, W2), dtype=np.float32)
    d2 = np.where(blk > 0.97, d2 * np.float32(1.3), d2).astype(np.float32)
    out["depth2"] = d2
    out["mask2"] = depth_mod.reliable_depth_mask_range(torch.from_numpy(d2)).numpy()
    np.savez_compressed(OUT / "g1_rays_mask.npz", **out)
    print("g1: mask true fraction", out["mask"].mean(), out["mask2"].mean())


def make_g3(cameras):
    """Ray tables of the two non-pinhole exo camera models, built the way scripts/render_vipe_pointcloud.py:384-398 does."""
    out = {}
    H, W = 20, 28
    for tag, s, intr in (("mei_a", 1, [14.0, 14.5, 13.6, 9.7, 0.8]), ("mei_b", 2, [11.0, 10.5, 14.0, 10.0, 1.35]),
                         ("mei_c", 3, [30.0, 30.0, 14.0, 10.0, 0.0])):
        model = cameras.CameraType.MEI.build_camera_model(torch.tensor(intr, dtype=torch.float32))
        dv, du = torch.meshgrid(torch.arange(H).float()[::s], torch.arange(W).float()[::s], indexing="ij")
        pts, _, _ = model.iproj_disp(torch.ones_like(dv), du, dv)
        rays = pts[..., :3].numpy()
        rays /= rays[..., 2:3]
        out[f"{tag}_intr"], out[f"{tag}_s"], out[tag] = np.asarray(intr, np.float32), np.int32(s), rays.astype(np.float32)
    for tag, s in (("pano_a", 1), ("pano_b", 2)):
        model = cameras.CameraType.PANORAMA.build_camera_model(torch.zeros(4))
        dv, du = torch.meshgrid(torch.arange(H).float()[::s], torch.arange(W).float()[::s], indexing="ij")
        dv, du = dv / (H - 1), du / (W - 1)
        pts, _, _ = model.iproj_disp(torch.ones_like(dv), du, dv)
        out[f"{tag}_s"], out[tag] = np.int32(s), pts[..., :3].numpy().astype(np.float32)
    out["H"], out["W"] = np.int32(H), np.int32(W)
    np.savez_compressed(OUT / "g3_rays_mei_pano.npz", **out)


def make_g2():
    rng = np.random.default_rng(1)
    out = {}
    for F in (1, 5, 16, 17, 49, 300):
        x = rng.random((F, 8, 16), dtype=np.float32) * 5
        x = (np.round(x * 255) / np.float32(255)).astype(np.float32)
        nanmask = rng.random((F, 8, 16)) < 0.3
        x[nanmask] = np.nan
        x[:, 0, 0] = np.nan  # never valid
        x[:, 1, 1] = np.float32(51.0 / 255.0)  # constant
        t = torch.from_numpy(x)
        y = torch.nan_to_num(torch.nanmean(t, dim=0), nan=0.0).numpy()
        out[f"x_{F}"] = x
        out[f"y_{F}"] = y
    np.savez_compressed(OUT / "g2_nanmean.npz", **out)


def make_g4():
    from PIL import Image
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    a[:48] = rng.integers(0, 80, (48, 128, 3), dtype=np.uint8)  # dark half: grey values around the threshold
    a[0, :3] = [[30, 30, 30], [29, 30, 31], [100, 0, 2]]
    L = np.asarray(Image.fromarray(a, "RGB").convert("L"))
    np.savez_compressed(OUT / "g4_pil_luma.npz", rgb=a, L=L)


def make_cameras():
    base = REF / "data_preprocess/example"
    d = json.load(open(base / "annotations/ego_pose/test/camera_pose/1410e2ce-a62c-4a2c-8453-6eb172a58965.json"))
    with open(base / "takes/cmu_bike08_2/trajectory/online_calibration.jsonl") as f:
        line0 = json.loads(f.readline())
    rgb = [c for c in line0["CameraCalibrations"] if c.get("Label") == "camera-rgb"][0]
    out = {
        "take_name": d["metadata"]["take_name"],
        "cam01": {"camera_intrinsics": d["cam01"]["camera_intrinsics"],
                  "camera_extrinsics": d["cam01"]["camera_extrinsics"]},
        "aria01": {"camera_intrinsics": d["aria01"]["camera_intrinsics"],
                   "camera_extrinsics": [d["aria01"]["camera_extrinsics"][str(i)] for i in range(128)]},
        "online_calibration_line0": {"CameraCalibrations": [{"Label": rgb["Label"],
                                                             "Projection": rgb["Projection"]}]},
    }
    json.dump(out, open(OUT / "cameras_cmu_bike08_2.json", "w"))


if __name__ == "__main__":
    cams, dep = load_reference_helpers()
    make_g1(cams, dep)
    make_g2()
    make_g3(cams)
    make_g4()
    make_cameras()
    print("golden fixtures written to", OUT)
