"""Device-side engine: owns HBM buffers (as torch tensors) and drives the C ABI on the current HIP stream.

torch is used for device memory, streams and (elsewhere) torch.distributed only; all arithmetic of the path runs in
the HIP kernels behind include/egoprior.h.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import Camera, RasterParams, UnprojectParams, check

F32 = np.float32


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class DeviceCloud:
    """A compacted cloud in HBM: `points` is (capacity, 4) float32 whose 4th column holds the packed colour bytes;
    `count` (int32[1]) is the live length, `flags` (int32[1]) is 1 when some colour byte exceeds 1."""
    points: torch.Tensor
    count: torch.Tensor
    flags: torch.Tensor
    xyz64: Optional[torch.Tensor] = None

    @property
    def capacity(self) -> int:
        return int(self.points.shape[0])

    def n(self) -> int:
        return int(self.count.item())

    def to_numpy(self):
        """(points float32 (N,3), colours uint8 (N,3)) of the live part."""
        n = self.n()
        p = self.points[:n].cpu().numpy()
        rgba = p[:, 3].copy().view(np.uint32)
        cols = np.stack([rgba & 0xFF, (rgba >> 8) & 0xFF, (rgba >> 16) & 0xFF], axis=1).astype(np.uint8)
        return p[:, :3].copy(), cols


# ----------------------------------------------------------------------------------------------------------------------
# host-side camera set-up (mirrors the reference's host code; tiny, float32 numpy)
# ----------------------------------------------------------------------------------------------------------------------
def radius_ndc(point_size, H: int, W: int) -> float:
    """scripts/render_vipe_pointcloud.py:853-855."""
    r = float(point_size) * (2.0 / max(W, H))
    return max(1e-5, min(0.25, r))


def make_raster_params(point_size, H: int, W: int, K: int = 20, rotate_cw: bool = False,
                       colors_gt1: int = -1) -> RasterParams:
    r = radius_ndc(point_size, H, W)
    return RasterParams(radius=float(F32(r)), radius_sq_w=float(F32(r * r)), K=int(K), rotate_cw=int(rotate_cw),
                        colors_gt1=int(colors_gt1))


def split_w2c(T_w2c):
    T = np.asarray(T_w2c)
    if T.shape == (4, 4):
        return T[:3, :3], T[:3, 3]
    if T.shape == (3, 4):
        return T[:, :3], T[:, 3]
    raise ValueError(f"T_w2c must be (3,4) or (4,4), got {T.shape}")


def pinhole_camera(R_cv, t_cv, K, H: int, W: int) -> Camera:
    """pytorch3d cameras_from_opencv_projection(R, tvec, camera_matrix, image_size=[[H, W]]) in float32
    (scripts/render_vipe_pointcloud.py:840-850)."""
    R = np.asarray(R_cv, dtype=F32)
    t = np.asarray(t_cv, dtype=F32)
    K = np.asarray(K, dtype=F32)
    scale = F32(min(W, H)) / F32(2.0)
    c0 = np.array([F32(W) / F32(2.0), F32(H) / F32(2.0)], dtype=F32)
    focal = (np.array([K[0, 0], K[1, 1]], dtype=F32) / scale).astype(F32)
    pp = (-(np.array([K[0, 2], K[1, 2]], dtype=F32) - c0) / scale).astype(F32)
    R_p3d = R.T.copy()
    R_p3d[:, :2] *= F32(-1)
    T_p3d = t.copy()
    T_p3d[:2] *= F32(-1)
    cam = Camera()
    cam.model, cam.H, cam.W = _lib.CAM_PINHOLE, int(H), int(W)
    cam.R[:] = [float(v) for v in R_p3d.reshape(-1)]
    cam.T[:] = [float(v) for v in T_p3d]
    cam.fx, cam.fy, cam.px, cam.py = float(focal[0]), float(focal[1]), float(pp[0]), float(pp[1])
    return cam


def fisheye_camera(R_cv, t_cv, f_px, cx_px, cy_px, radial, tangential, thin_prism, H: int, W: int) -> Camera:
    """The reference's FishEyeCameras set-up (scripts/render_vipe_pointcloud.py:990-1062)."""
    R = np.asarray(R_cv, dtype=F32)
    t = np.asarray(t_cv, dtype=F32)
    Cm = (np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]], dtype=F32)
          @ np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]], dtype=F32))
    R_p3d = (R.T @ Cm).astype(F32)
    T_p3d = (Cm.T @ t).astype(F32)
    f_ndc = 2.0 * f_px / W
    cx_ndc = 2.0 * (cx_px / W) - 1.0
    cy_ndc = 1.0 - 2.0 * (cy_px / H)
    cam = Camera()
    cam.model, cam.H, cam.W = _lib.CAM_FISHEYE, int(H), int(W)
    cam.R[:] = [float(v) for v in R_p3d.reshape(-1)]
    cam.T[:] = [float(v) for v in T_p3d]
    cam.fx = cam.fy = float(F32(f_ndc))
    cam.px, cam.py = float(F32(cx_ndc)), float(F32(cy_ndc))
    cam.radial[:] = [float(F32(v)) for v in radial]
    cam.tangential[:] = [float(F32(v)) for v in tangential]
    cam.thin_prism[:] = [float(F32(v)) for v in thin_prism]
    return cam


def make_unproject_params(H, W, subsample, select, intr, c2w, T_cam_to_world) -> UnprojectParams:
    p = UnprojectParams()
    p.H, p.W, p.subsample, p.select = int(H), int(W), int(subsample), int(select)
    intr = np.asarray(intr, dtype=F32)
    p.fx, p.fy, p.cx, p.cy = float(intr[0]), float(intr[1]), float(intr[2]), float(intr[3])
    c = np.asarray(c2w, dtype=F32)
    p.c2w[:] = [float(v) for v in c[:3, :4].reshape(-1)]
    T = np.asarray(T_cam_to_world, dtype=np.float64)
    p.T[:] = [float(v) for v in T[:3, :4].reshape(-1)]
    return p


# ----------------------------------------------------------------------------------------------------------------------
# engine
# ----------------------------------------------------------------------------------------------------------------------
class Engine:
    """Holds reusable workspaces on one GPU.  Not thread-safe; one per process/GPU."""

    def __init__(self, device: str | torch.device = "cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("the ego-prior renderer needs a ROCm GPU (there is no CPU path)")
        self.L = _lib.lib()
        self._ws = {}

    # -- buffers ------------------------------------------------------------------------------------------------------
    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        cur = self._ws.get(key)
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(int(nbytes * 1.25) + 256 if cur is not None else int(nbytes), dtype=torch.uint8,
                              device=self.device)
            self._ws[key] = cur
        return cur

    def new_cloud(self, capacity: int, want_xyz64: bool = False) -> DeviceCloud:
        cap = max(int(capacity), 1)
        return DeviceCloud(
            points=torch.empty((cap, 4), dtype=torch.float32, device=self.device),
            count=torch.zeros(1, dtype=torch.int32, device=self.device),
            flags=torch.zeros(1, dtype=torch.int32, device=self.device),
            xyz64=torch.empty((cap, 3), dtype=torch.float64, device=self.device) if want_xyz64 else None)

    def depth_to_device(self, depth) -> torch.Tensor:
        """Depth maps stay float16 if they arrive as float16 (the precision the artifacts are stored in): the kernels
        widen them on the fly, which is exact.  Anything else becomes float32."""
        is_f16 = (depth.dtype == torch.float16) if isinstance(depth, torch.Tensor) else (np.asarray(depth).dtype == np.float16)
        return self.to_device(depth, torch.float16 if is_f16 else torch.float32)

    def _by_depth(self, name: str, d: torch.Tensor):
        return getattr(self.L, name + "_f16") if d.dtype == torch.float16 else getattr(self.L, name)

    def to_device(self, a, dtype) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:
            a = a.copy()
        return torch.from_numpy(a).to(device=self.device, dtype=dtype).contiguous()

    # -- R5 -----------------------------------------------------------------------------------------------------------
    def reliable_depth_mask(self, depth) -> torch.Tensor:
        d = self.depth_to_device(depth)
        H, W = d.shape
        out = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._by_depth("egoprior_reliable_depth_mask", d)(_ptr(d), H, W, _ptr(out), _stream()))
        return out.bool()

    # -- R6 / R7 ------------------------------------------------------------------------------------------------------
    def unproject_frame(self, depth, rgb, mask, intr, c2w, T_cam_to_world, subsample: int = 2,
                        select: int = _lib.SELECT_STATIC, out: Optional[DeviceCloud] = None,
                        want_xyz64: bool = False, rays=None) -> DeviceCloud:
        """rays: (ceil(H/s), ceil(W/s), 3) float32 ray table of a non-pinhole exo camera (egoprior.cameras.ray_table);
        None = pinhole rays from `intr`."""
        d = self.depth_to_device(depth)
        c = self.to_device(rgb, torch.uint8)
        m = None if mask is None else self.to_device(mask, torch.uint8)
        H, W = d.shape
        if tuple(c.shape) != (H, W, 3):
            raise ValueError(f"rgb must be ({H},{W},3), got {tuple(c.shape)}")
        if m is not None and tuple(m.shape) != (H, W):
            raise ValueError(f"mask must be ({H},{W}), got {tuple(m.shape)}")
        s = int(subsample)
        ns = ((H + s - 1) // s) * ((W + s - 1) // s)
        if out is None:
            out = self.new_cloud(ns, want_xyz64)
        if out.capacity < ns or (want_xyz64 and out.xyz64 is None):
            raise ValueError("output cloud too small")
        p = make_unproject_params(H, W, s, select, intr, c2w, T_cam_to_world)
        nbytes = self.L.egoprior_unproject_workspace_bytes(H, W, s)
        ws = self._workspace("unproject", nbytes)
        with torch.cuda.device(self.device):
            if rays is not None:
                r = self.to_device(rays, torch.float32)
                if tuple(r.shape) != ((H + s - 1) // s, (W + s - 1) // s, 3):
                    raise ValueError("ray table does not match the subsampled frame")
                check(self.L.egoprior_unproject_frame_rays(
                    C.byref(p), _ptr(d), int(d.dtype == torch.float16), _ptr(r), _ptr(c), _ptr(m), _ptr(out.points),
                    _ptr(out.xyz64) if want_xyz64 else None, _ptr(out.count), _ptr(out.flags), _ptr(ws), ws.numel(),
                    _stream()))
            else:
                check(self._by_depth("egoprior_unproject_frame", d)(
                    C.byref(p), _ptr(d), _ptr(c), _ptr(m), _ptr(out.points), _ptr(out.xyz64) if want_xyz64 else None,
                    _ptr(out.count), _ptr(out.flags), _ptr(ws), ws.numel(), _stream()))
        return out

    # -- render-entry filters -----------------------------------------------------------------------------------------
    def filter_cloud(self, xyz, rgb_u8) -> DeviceCloud:
        x = self.to_device(xyz, torch.float32).reshape(-1, 3)
        c = self.to_device(rgb_u8, torch.uint8).reshape(-1, 3)
        if x.shape[0] != c.shape[0]:
            raise ValueError("points and colours differ in length")
        n = int(x.shape[0])
        out = self.new_cloud(n)
        nbytes = self.L.egoprior_filter_workspace_bytes(n)
        ws = self._workspace("filter", nbytes)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_filter_cloud(_ptr(x) if n else None, _ptr(c) if n else None, n, _ptr(out.points),
                                               _ptr(out.count), _ptr(out.flags), _ptr(ws), ws.numel(), _stream()))
        return out

    # -- R8 -----------------------------------------------------------------------------------------------------------
    def meanbg_new_state(self, H: int, W: int) -> torch.Tensor:
        return torch.zeros((17, H, W), dtype=torch.float32, device=self.device)

    def meanbg_accumulate(self, state: torch.Tensor, depth, rgb, mask, first_index: int = 0) -> None:
        d = self.depth_to_device(depth)
        c = self.to_device(rgb, torch.uint8)
        m = None if mask is None else self.to_device(mask, torch.uint8)
        if d.dim() == 2:
            d, c = d[None], c[None]
            m = None if m is None else m[None]
        Fn, H, W = d.shape
        if tuple(state.shape) != (17, H, W):
            raise ValueError("state does not match the frame size")
        with torch.cuda.device(self.device):
            check(self._by_depth("egoprior_meanbg_accumulate", d)(_ptr(state), _ptr(d), _ptr(c), _ptr(m), Fn,
                                                                  int(first_index), H, W, _stream()))

    def meanbg_finalize(self, state: torch.Tensor):
        _, H, W = state.shape
        md = torch.empty((H, W), dtype=torch.float32, device=self.device)
        mc = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_meanbg_finalize(_ptr(state), H, W, _ptr(md), _ptr(mc), _stream()))
        return md, mc

    def meanbg_finalize_f32(self, state: torch.Tensor, subsample: int = 1):
        """The viewer's flavour (vipe/utils/viser.py:189-214): float32 means of the [::s, ::s] pixels, colours unquantised."""
        _, H, W = state.shape
        s = int(subsample)
        Hs, Ws = (H + s - 1) // s, (W + s - 1) // s
        md = torch.empty((Hs, Ws), dtype=torch.float32, device=self.device)
        mc = torch.empty((Hs, Ws, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_meanbg_finalize_f32(_ptr(state), H, W, s, _ptr(md), _ptr(mc), _stream()))
        return md, mc

    def meanbg_export(self, state: torch.Tensor):
        _, H, W = state.shape
        sums = torch.empty((4, H, W), dtype=torch.float32, device=self.device)
        cnt = torch.empty((H, W), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_meanbg_export(_ptr(state), H, W, _ptr(sums), _ptr(cnt), _stream()))
        return sums, cnt

    def meanbg_import(self, state: torch.Tensor, sums: torch.Tensor, cnt: torch.Tensor) -> None:
        _, H, W = state.shape
        with torch.cuda.device(self.device):
            check(self.L.egoprior_meanbg_import(_ptr(state), H, W, _ptr(sums.contiguous()), _ptr(cnt.contiguous()),
                                                _stream()))

    # -- R9..R12 ------------------------------------------------------------------------------------------------------
    def project_points(self, cam: Camera, a: DeviceCloud, b: Optional[DeviceCloud] = None) -> torch.Tensor:
        n = a.capacity + (b.capacity if b is not None else 0)
        out = torch.empty((max(n, 1), 4), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_project_points(C.byref(cam), _ptr(a.points), a.capacity, _ptr(a.count),
                                                 _ptr(b.points) if b is not None else None,
                                                 b.capacity if b is not None else 0,
                                                 _ptr(b.count) if b is not None else None, _ptr(out), _stream()))
        return out

    def _alloc_outputs(self, H, W, K, rotate_cw, want_f32, want_fragments):
        shape = (W, H, 3) if rotate_cw else (H, W, 3)
        img = torch.empty(shape, dtype=torch.uint8, device=self.device)
        f32 = torch.empty((H, W, 3), dtype=torch.float32, device=self.device) if want_f32 else None
        idx = zb = ds = None
        if want_fragments:
            idx = torch.empty((H, W, K), dtype=torch.int32, device=self.device)
            zb = torch.empty((H, W, K), dtype=torch.float32, device=self.device)
            ds = torch.empty((H, W, K), dtype=torch.float32, device=self.device)
        return img, f32, idx, zb, ds

    def rasterize_composite(self, ndc: torch.Tensor, n_live: Optional[torch.Tensor], rp: RasterParams, H: int, W: int,
                            flags: Sequence[Optional[torch.Tensor]] = (None, None), want_f32: bool = False,
                            want_fragments: bool = False):
        """ndc: (n,4) float32 device tensor (x_ndc, y_ndc, z, packed colour)."""
        n = int(ndc.shape[0])
        img, f32, idx, zb, ds = self._alloc_outputs(H, W, rp.K, bool(rp.rotate_cw), want_f32, want_fragments)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        nbytes = self.L.egoprior_raster_workspace_bytes(H, W, n, rp.radius)
        ws = self._workspace("raster", nbytes)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_rasterize_composite(C.byref(rp), H, W, _ptr(ndc), n, _ptr(n_live), _ptr(flags[0]),
                                                      _ptr(flags[1]), _ptr(img), _ptr(f32), _ptr(idx), _ptr(zb),
                                                      _ptr(ds), _ptr(status), _ptr(ws), ws.numel(), _stream()))
        return dict(image=img, image_f32=f32, idx=idx, zbuf=zb, dists=ds, status=status)

    # -- output side ---------------------------------------------------------------------------------------------------
    def white_pixel_counts(self, frames, threshold: float = 30.0, want_masks: bool = False, grey: str = "pil"):
        """Per-frame count of pixels whose grey value reaches `threshold`
        (data_preprocess/scripts/postprocess_render_dir.py:17-31, 46-67).  grey="pil": PIL's Image.convert("L") integer
        luma, the branch the script takes whenever PIL imports; grey="float": its imageio fallback,
        float32(0.299 r + 0.587 g + 0.114 b).  frames: (n, H, W, 3) uint8 (device tensor or array).  Returns the int32
        device tensor of counts, and the (n, H, W) 0/255 masks if asked for."""
        rule = {"pil": _lib.GREY_PIL_L, "float": _lib.GREY_FLOAT}[grey]
        fr = self.to_device(frames, torch.uint8)
        if fr.dim() == 3:
            fr = fr[None]
        if fr.dim() != 4 or fr.shape[-1] != 3:
            raise ValueError("frames must be (n, H, W, 3) uint8")
        n, H, W, _ = fr.shape
        counts = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        masks = torch.empty((n, H, W), dtype=torch.uint8, device=self.device) if want_masks else None
        with torch.cuda.device(self.device):
            check(self.L.egoprior_white_pixel_counts(_ptr(fr), n, H, W, float(threshold), rule, _ptr(counts),
                                                     _ptr(masks), _stream()))
        return (counts[:n], masks) if want_masks else counts[:n]

    def postprocess_background(self, image_f32, background_f32, rotate_cw: bool) -> torch.Tensor:
        """uint8 frame from the compositor's float32 output and a float32 background image in 0..1 (the reference's
        clamp / dark-pixel / background / uint8 / rotation steps, scripts/render_vipe_pointcloud.py:873-888)."""
        img = self.to_device(image_f32, torch.float32)
        bg = self.to_device(background_f32, torch.float32)
        H, W = int(img.shape[0]), int(img.shape[1])
        if tuple(img.shape) != (H, W, 3) or tuple(bg.shape) != (H, W, 3):
            raise ValueError("image and background must both be (H, W, 3)")
        out = torch.empty((W, H, 3) if rotate_cw else (H, W, 3), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_postprocess_background(_ptr(img), _ptr(bg), H, W, int(bool(rotate_cw)), _ptr(out),
                                                         _stream()))
        return out

    def render_frame(self, cam: Camera, rp: RasterParams, a: DeviceCloud, b: Optional[DeviceCloud] = None,
                     want_f32: bool = False, want_fragments: bool = False, out_image: Optional[torch.Tensor] = None):
        H, W = cam.H, cam.W
        n = a.capacity + (b.capacity if b is not None else 0)
        img, f32, idx, zb, ds = self._alloc_outputs(H, W, rp.K, bool(rp.rotate_cw), want_f32, want_fragments)
        if out_image is not None:
            if out_image.shape != img.shape or out_image.dtype != torch.uint8 or not out_image.is_contiguous():
                raise ValueError("out_image has the wrong shape/dtype")
            img = out_image
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        nbytes = self.L.egoprior_render_workspace_bytes(H, W, n, rp.radius)
        ws = self._workspace("render", nbytes)
        with torch.cuda.device(self.device):
            check(self.L.egoprior_render_frame(
                C.byref(cam), C.byref(rp), _ptr(a.points), a.capacity, _ptr(a.count), _ptr(a.flags),
                _ptr(b.points) if b is not None else None, b.capacity if b is not None else 0,
                _ptr(b.count) if b is not None else None, _ptr(b.flags) if b is not None else None,
                _ptr(img), _ptr(f32), _ptr(idx), _ptr(zb), _ptr(ds), _ptr(status), _ptr(ws), ws.numel(), _stream()))
        return dict(image=img, image_f32=f32, idx=idx, zbuf=zb, dists=ds, status=status)


# ----------------------------------------------------------------------------------------------------------------------
# frames in flight
# ----------------------------------------------------------------------------------------------------------------------
def hardware_queues() -> int:
    """Hardware queues HIP gives this process (GPU_MAX_HW_QUEUES, 4 by default)."""
    import os
    try:
        return max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        return 4


def check_stream_budget(n_slots: int, extra_priorities: Sequence[int] = (), oversubscribe: bool = False) -> None:
    """Raises ValueError for a frame pipeline whose streams would share hardware queues (more slots than queues) or mix
    priorities, the two ingredients of the stall described in FramePipeline's docstring."""
    if oversubscribe:
        return
    hwq = hardware_queues()
    if n_slots > hwq:
        raise ValueError(f"{n_slots} frames in flight need {n_slots} streams, but this process has {hwq} hardware queues "
                         "(GPU_MAX_HW_QUEUES): streams that share a queue serialise, event waits included; pass "
                         "oversubscribe=True to try it anyway")
    if any(int(p) != 0 for p in extra_priorities):
        raise ValueError("the frame pipeline takes default-priority streams only (pass oversubscribe=True to override)")


class FramePipeline:
    """Renders a sequence of ego frames with several frames in flight on one GPU.

    The frames of the reference's loop (scripts/render_vipe_pointcloud.py:1178-1232) do not depend on one another: each
    takes the shared static cloud, its own exo frame's dynamic points and its own ego pose.  Consecutive frames are
    therefore issued round-robin to `slots` HIP streams, each slot with its own dynamic-point buffer, workspaces and
    status word, so the short single-workgroup scans and the tail of one frame's splat overlap the next frame's
    front end.  Results are the same bits as rendering the frames one after the other.

        pipe = FramePipeline(engine, static_cloud, rp, He, We, exo_hw=(H, W), subsample=s)
        for f in frames:
            pipe.submit(cams[f], out_u8[f], up=unproject_params, depth=depth[f], rgb=rgb[f], mask=mask[f])
        pipe.finish()          # joins the streams; raises if a tile list overflowed

    Inputs and outputs must stay alive and untouched until finish().  Not thread-safe.

    Streams and hardware queues.  A process has GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says
    otherwise) and HIP multiplexes its streams onto them, so two streams that share a queue run strictly one packet after
    the other, event-wait packets included.  The pipeline therefore keeps one default-priority stream per slot, lets
    slots wait only on the main stream (fork, `after_main`) and the main stream on the slots (join), and refuses more
    slots than hardware queues or streams of another priority unless `oversubscribe=True`: an experiment that gave each
    slot a second, high-priority stream for its binning kernels (8 slot streams + the default stream on 4 queues, with
    event waits from a slot's splat stream on its binning stream and back, for the reuse of the slot's workspace) stopped
    making progress in its warm-up and had to be killed, and more than 4 default-priority slots measured slower than 4.
    """

    def __init__(self, engine: Engine, static: DeviceCloud, rp: RasterParams, He: int, We: int,
                 exo_hw: Optional[Sequence[int]] = None, subsample: int = 1, slots: int = 4,
                 extra_streams: Optional[Sequence["torch.cuda.Stream"]] = None, oversubscribe: bool = False):
        self.E, self.L = engine, engine.L
        self.static, self.rp, self.He, self.We = static, rp, int(He), int(We)
        self.n_slots = max(1, int(slots))
        check_stream_budget(self.n_slots, [getattr(st, "priority", 0) for st in (extra_streams or [])], oversubscribe)
        dev = engine.device
        self.status = torch.zeros(self.n_slots, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            self.main = torch.cuda.current_stream()
            # slot 0 runs on the stream that was current at construction; the others get streams of their own
            # (extra_streams: the caller's choice of streams for slots 1.., e.g. to keep them off a hardware queue that
            # a communication library's stream already occupies)
            extra = list(extra_streams) if extra_streams is not None else \
                [torch.cuda.Stream(device=dev) for _ in range(self.n_slots - 1)]
            if len(extra) != self.n_slots - 1:
                raise ValueError("extra_streams must hold slots - 1 streams")
            self.streams = [self.main] + extra
        self.exo = None
        self.dyn_cap = 0
        self.ws_u = [None] * self.n_slots
        self.dyns = [None] * self.n_slots
        self.ws_r = None
        self._size_buffers(None if exo_hw is None else (int(exo_hw[0]), int(exo_hw[1]), int(subsample)))
        self._h = [C.c_void_p(st.cuda_stream) for st in self.streams]
        self._next = 0
        self._forked = False
        shape = (self.We, self.He, 3) if rp.rotate_cw else (self.He, self.We, 3)
        self.image_shape = shape

    def _size_buffers(self, exo) -> None:
        """(Re)allocates the per-slot buffers for exo frames of size exo = (H, W, subsample), or for none."""
        dev = self.E.device
        if exo is not None:
            H, W, s = exo
            self.dyn_cap = ((H + s - 1) // s) * ((W + s - 1) // s)
            nb = self.L.egoprior_unproject_workspace_bytes(H, W, s)
            self.ws_u = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(self.n_slots)]
            self.dyns = [self.E.new_cloud(self.dyn_cap) for _ in range(self.n_slots)]
            self.exo = exo
        nb = self.L.egoprior_render_workspace_bytes(self.He, self.We, self.static.capacity + self.dyn_cap,
                                                    self.rp.radius)
        self.ws_r = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(self.n_slots)]

    def _fork(self):
        if not self._forked:  # everything queued on the main stream so far happens before the slots' work
            for st in self.streams[1:]:
                st.wait_stream(self.main)
            self._forked = True

    def submit(self, cam: Camera, out_image: torch.Tensor, up: Optional[UnprojectParams] = None,
               depth: Optional[torch.Tensor] = None, rgb: Optional[torch.Tensor] = None,
               mask: Optional[torch.Tensor] = None, slot: Optional[int] = None, after_main: bool = False,
               rays: Optional[torch.Tensor] = None) -> int:
        """Queues one frame; returns the slot it went to.  With `up`, `depth`, `rgb` (and `mask`) given, the frame's
        dynamic points are extracted first (same slot, same stream) and rendered together with the static cloud.
        `after_main`: the frame's inputs were produced by work queued on the main stream since the last submit (an
        upload, say), so the slot's stream waits for the main stream first."""
        if cam.H != self.He or cam.W != self.We:
            raise ValueError("camera size differs from the pipeline's")
        if tuple(out_image.shape) != tuple(self.image_shape) or out_image.dtype != torch.uint8 \
                or not out_image.is_contiguous():
            raise ValueError(f"out_image must be a contiguous uint8 tensor of shape {tuple(self.image_shape)}")
        k = self._next % self.n_slots if slot is None else int(slot)
        if slot is None:
            self._next += 1
        with torch.cuda.device(self.E.device):
            if after_main and k > 0:
                self.streams[k].wait_stream(self.main)
            return self._submit(k, cam, out_image, up, depth, rgb, mask, rays)

    def _submit(self, k, cam, out_image, up, depth, rgb, mask, rays=None) -> int:
        self._fork()
        d = None
        if up is not None:
            if self.exo is None:  # first exo frame of a pipeline built without exo_hw: drain, then size the buffers
                self.join()
                self.main.synchronize()
                self._size_buffers((int(up.H), int(up.W), int(up.subsample)))
                self._fork()
            H, W, s = self.exo
            if (up.H, up.W, up.subsample) != (H, W, s) or tuple(depth.shape) != (H, W) \
                    or tuple(rgb.shape) != (H, W, 3) or (mask is not None and tuple(mask.shape) != (H, W)):
                raise ValueError("exo frame does not match the pipeline's exo_hw / subsample")
            d = self.dyns[k]
            if depth.dtype not in (torch.float32, torch.float16) or not depth.is_contiguous():
                raise ValueError("depth must be a contiguous float32 or float16 tensor")
            if rays is not None:  # device ray table of a non-pinhole exo camera, (ceil(H/s), ceil(W/s), 3) float32
                check(self.L.egoprior_unproject_frame_rays(
                    C.byref(up), _ptr(depth), int(depth.dtype == torch.float16), _ptr(rays), _ptr(rgb), _ptr(mask),
                    _ptr(d.points), None, _ptr(d.count), _ptr(d.flags), _ptr(self.ws_u[k]), self.ws_u[k].numel(),
                    self._h[k]))
            else:
                check(self.E._by_depth("egoprior_unproject_frame", depth)(
                    C.byref(up), _ptr(depth), _ptr(rgb), _ptr(mask), _ptr(d.points), None, _ptr(d.count),
                    _ptr(d.flags), _ptr(self.ws_u[k]), self.ws_u[k].numel(), self._h[k]))
        a = self.static
        check(self.L.egoprior_render_frame(
            C.byref(cam), C.byref(self.rp), _ptr(a.points), a.capacity, _ptr(a.count), _ptr(a.flags),
            _ptr(d.points) if d is not None else None, d.capacity if d is not None else 0,
            _ptr(d.count) if d is not None else None, _ptr(d.flags) if d is not None else None,
            _ptr(out_image), None, None, None, None, C.c_void_p(self.status.data_ptr() + 4 * k),
            _ptr(self.ws_r[k]), self.ws_r[k].numel(), self._h[k]))
        return k

    def join(self) -> None:
        """The main stream continues after everything queued on the slots' streams (no host synchronisation)."""
        for st in self.streams[1:]:
            self.main.wait_stream(st)
        self._forked = False

    def finish(self) -> None:
        self.join()
        self.main.synchronize()
        # the kernels only ever raise a slot's status word, so one word covers every frame the slot has rendered
        if int(self.status.abs().sum().item()) != 0:
            self.status.zero_()
            raise RuntimeError("tile lists overflowed their workspace (status != 0)")
