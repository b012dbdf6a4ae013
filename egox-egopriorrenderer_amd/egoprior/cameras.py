"""Ray tables of the exo camera models ViPE can emit (vipe/utils/cameras.py), on the pixel grid the render script uses
(scripts/render_vipe_pointcloud.py:384-398).  A table is built once per clip on the host, with the same torch operations
the reference's camera classes apply, and handed to the unprojection kernel; the pinhole model needs no table (the kernel
evaluates (u - cx) / fx itself)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

CAMERA_TYPES = ("PINHOLE", "MEI", "PANORAMA")


def ray_table(camera_type: str, intr, H: int, W: int, s: int) -> Optional[np.ndarray]:
    """(ceil(H/s), ceil(W/s), 3) float32 rays for MEI / PANORAMA, None for PINHOLE.

    MEI (unified model, intrinsics fx, fy, cx, cy, k1; cameras.py:228-250), then divided by z (= 1) like the script
    does for every model but the panorama.  PANORAMA (equirectangular; cameras.py:366-387): pixel coordinates are
    first normalised by (W-1), (H-1); the rays are unit vectors and are not divided by z."""
    ct = str(camera_type).upper()
    if ct == "PINHOLE":
        return None
    dv, du = torch.meshgrid(torch.arange(H).float()[::s], torch.arange(W).float()[::s], indexing="ij")
    if ct == "MEI":
        it = torch.as_tensor(np.asarray(intr, dtype=np.float32))
        if it.numel() < 5:
            raise ValueError("the MEI camera model takes 5 intrinsics (fx, fy, cx, cy, k1)")
        fx, fy, cx, cy, k1 = it.view(1, 1, -1)[..., :5].unbind(dim=-1)
        u_bar = (du - cx) / fx
        v_bar = (dv - cy) / fy
        r2 = u_bar ** 2 + v_bar ** 2
        q = torch.sqrt(1 + (1 - k1 ** 2) * r2)
        factor = (k1 + q) / (1 + r2)
        X = u_bar * factor / (factor - k1)
        Y = v_bar * factor / (factor - k1)
        rays = torch.stack([X, Y, torch.ones_like(X)], dim=-1).numpy()
        rays /= rays[..., 2:3]
        return np.ascontiguousarray(rays, dtype=np.float32)
    if ct == "PANORAMA":
        dv = dv / (H - 1)
        du = du / (W - 1)
        theta = (du - 0.5) * 2 * torch.pi
        phi = dv * torch.pi
        sin_phi = torch.sin(phi)
        rays = torch.stack((sin_phi * torch.sin(theta), -torch.cos(phi), sin_phi * torch.cos(theta)), dim=-1)
        return np.ascontiguousarray(rays.numpy(), dtype=np.float32)
    raise ValueError(f"Un-implemented camera type: {camera_type}")  # cameras.py:42
