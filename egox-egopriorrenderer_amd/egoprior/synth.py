"""Deterministic synthetic ViPE-format sequences (no dataset or network needed).

Scene, in the exo camera frame (OpenCV axes: x right, y down, z forward): a ground plane, a back wall, two side
walls, and a sphere (instance id 1) moving on a circle.  Depth is ray-cast analytically, then rounded to float16 the way
ViPE stores it (EXR HALF, vipe/utils/io.py:274-277) and returned as float32.  Smoothness matters: the 5x5 reliability
test rejects anything whose local range/mean is >= 0.1.  Colours depend only on the hit surface point (checker + hashed
noise), so the static background is constant over time.  The exo pose is the identity for every frame (fixed camera).

Everything is torch and device-agnostic: the tests build small clips on the CPU, bench.py builds the full-size clip
directly in HBM.
"""
from __future__ import annotations

import json
import math
from pathlib import Path

import numpy as np
import torch

GROUND_Y = 1.0
BACK_Z = 6.0
LEFT_X = -3.0
RIGHT_X = 3.5
SPHERE_R = 0.4


def exo_intrinsics(H: int, W: int) -> np.ndarray:
    """[fx, fy, cx, cy] of the synthetic exo camera."""
    return np.array([0.75 * W, 0.75 * W, W / 2.0, H / 2.0], dtype=np.float32)


def _hash01(ix: torch.Tensor, iy: torch.Tensor, seed: int) -> torch.Tensor:
    h = (ix.to(torch.int64) * 374761393 + iy.to(torch.int64) * 668265263 + seed * 982451653) & 0xFFFFFFFF
    h = ((h ^ (h >> 13)) * 1274126177) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return (h & 0xFFFF).to(torch.float32) / 65535.0


def sphere_centre(f: int, F: int):
    th = 2.0 * math.pi * (f / max(F, 1))
    return 1.2 * math.cos(th), 0.2, 3.5 + 1.0 * math.sin(th)


def make_frames(H: int, W: int, frames, F_total: int, device="cpu", seed: int = 0):
    """depth (n,H,W) float32, rgb (n,H,W,3) uint8, mask (n,H,W) uint8 for the given frame numbers."""
    device = torch.device(device)
    fx, fy, cx, cy = [float(v) for v in exo_intrinsics(H, W)]
    v, u = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                          torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    dx = (u - cx) / fx
    dy = (v - cy) / fy
    inf = torch.full_like(dx, float("inf"))
    # static surfaces: z-depth of the first hit (ray = (dx, dy, 1) * z)
    z_ground = torch.where(dy > 1e-6, GROUND_Y / dy.clamp_min(1e-6), inf)
    z_left = torch.where(dx < -1e-6, LEFT_X / dx.clamp_max(-1e-6), inf)
    z_right = torch.where(dx > 1e-6, RIGHT_X / dx.clamp_min(1e-6), inf)
    z_back = torch.full_like(dx, BACK_Z)
    zs = torch.stack([z_ground, z_left, z_right, z_back], dim=0)
    z_static, surf = zs.min(dim=0)
    px, py, pz = dx * z_static, dy * z_static, z_static
    chk = (torch.floor(px * 2.0) + torch.floor(py * 2.0) + torch.floor(pz * 2.0 + 0.25)).to(torch.int64) & 1
    base = torch.tensor([[0.55, 0.50, 0.42], [0.35, 0.45, 0.60], [0.60, 0.40, 0.35], [0.45, 0.58, 0.45]],
                        device=device)[surf]
    noise = _hash01(u, v, seed)
    shade = (0.75 + 0.25 * chk.to(torch.float32)) * (0.85 + 0.15 * noise)
    rgb_static = (base * shade[..., None]).clamp(0, 1)
    dd = dx * dx + dy * dy + 1.0
    depths, rgbs, masks = [], [], []
    for f in frames:
        sx, sy, sz = sphere_centre(int(f), F_total)
        dc = dx * sx + dy * sy + sz
        disc = dc * dc - dd * (sx * sx + sy * sy + sz * sz - SPHERE_R * SPHERE_R)
        t = (dc - torch.sqrt(disc.clamp_min(0))) / dd
        hit = (disc > 0) & (t > 0) & (t < z_static)
        depth = torch.where(hit, t, z_static)
        nx = (dx * t - sx) / SPHERE_R
        ny = (dy * t - sy) / SPHERE_R
        lam = (0.35 - 0.5 * ny + 0.3 * nx).clamp(0.15, 1.0)
        rgb_s = torch.stack([0.9 * lam, 0.25 * lam, 0.2 * lam], dim=-1)
        rgb = torch.where(hit[..., None], rgb_s, rgb_static)
        depths.append(depth.to(torch.float16).to(torch.float32))
        rgbs.append((rgb * 255.0).to(torch.uint8))
        masks.append(hit.to(torch.uint8))
    return torch.stack(depths), torch.stack(rgbs), torch.stack(masks)


# ----------------------------------------------------------------------------------------------------------------------
# cameras
# ----------------------------------------------------------------------------------------------------------------------
_FIXTURE = Path(__file__).resolve().parents[2] / "tests" / "golden" / "cameras_cmu_bike08_2.json"


def load_camera_fixture(path: Path | str = _FIXTURE) -> dict:
    """Camera parameters of Ego-Exo4D take cmu_bike08_2 (exo cam01 K + W2C, aria01 K + 128 per-frame W2C, and the
    camera-rgb FisheyeRadTanThinPrism parameters of the first online-calibration record)."""
    with open(path) as f:
        return json.load(f)


def exo_w2c(fixture: dict | None = None) -> np.ndarray:
    fx = load_camera_fixture() if fixture is None else fixture
    T = np.eye(4, dtype=np.float32)
    T[:3, :] = np.asarray(fx["cam01"]["camera_extrinsics"], dtype=np.float32)
    return T


def look_at_w2c(eye, target, up=(0.0, -1.0, 0.0)) -> np.ndarray:
    """OpenCV world->camera 4x4 (x right, y down, z forward) for a camera at `eye` looking at `target`."""
    eye = np.asarray(eye, dtype=np.float64)
    z = np.asarray(target, dtype=np.float64) - eye
    z /= np.linalg.norm(z)
    x = np.cross(-np.asarray(up, dtype=np.float64), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=0)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = -R @ eye
    return T


def ego_orbit_w2c(F: int, exo_world_to_cam: np.ndarray) -> list[np.ndarray]:
    """A head-like trajectory near the exo camera, looking into the scene; returned as world->ego 4x4 float32 in the
    same world frame the exo extrinsics live in (ego_w2c = ego_from_exo @ exo_w2c)."""
    out = []
    E = np.asarray(exo_world_to_cam, dtype=np.float64)
    for f in range(F):
        ph = 2.0 * math.pi * f / max(F, 1)
        eye = (0.35 * math.sin(ph), -0.25 + 0.10 * math.cos(2 * ph), -0.4 + 0.2 * math.cos(ph))
        tgt = (0.4 * math.sin(ph + 0.7), 0.35, 4.0)
        ego_from_exo = look_at_w2c(eye, tgt)
        out.append((ego_from_exo @ E).astype(np.float32))
    return out


def ego_pinhole_K(H_e: int, W_e: int) -> np.ndarray:
    return np.array([[0.6 * W_e, 0, W_e / 2.0], [0, 0.6 * W_e, H_e / 2.0], [0, 0, 1]], dtype=np.float32)
