"""Drop-in for the reference's scripts/render_vipe_pointcloud.py: same module-level function names, positional
order, defaults, return types (numpy in / numpy uint8 out) and CLI flags; the arithmetic runs in the HIP kernels behind
include/egoprior.h on an MI355X.  There is no CPU path: without a ROCm GPU the render/build functions raise.

Function-by-function correspondence (reference line numbers):
  generate_background ................. :39-72      scale_intrinsics ..................... :74-125
  load_aria_distortion_coeffs ......... :127-234    load_camera_params_from_meta ......... :237-301
  build_background_pointcloud ......... :304-456    compute_robust_mean_tensors .......... :459-473
  build_mean_background_pointcloud .... :477-656    build_dynamic_points_for_frame ....... :658-751
  render_points_pytorch3d ............. :754-890    render_points_fisheye ................ :892-1102
  project_points_to_image_sequential .. :1104-1237  get_parser / main and helpers ........ :1239-1453

Deliberate differences, all invisible in the outputs:
  * artifacts are decoded once, sequentially (the reference re-walks every iterator from frame 0 for every rendered
    frame, :682-692);
  * the static cloud is uploaded and filtered once per clip instead of once per frame (:800-832);
  * main() accepts the 4-tuple build_background_pointcloud returns (the reference's main unpacks 3 names from it and
    raises, :449 vs :1378), so the non-mean-background CLI path works;
  * colours must be integers in 0..255 (what every builder here and in the reference produces).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from . import engine as _eng
from .cameras import ray_table
from .io import (ArtifactPath, read_depth_artifacts, read_instance_artifacts, read_intrinsics_artifacts,
                 read_pose_artifacts, read_rgb_artifacts, write_video)

logger = logging.getLogger(__name__)

_ENGINE = None


def _engine():
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = _eng.Engine("cuda:0")
    return _ENGINE


# ----------------------------------------------------------------------------------------------------------------------
# small host-side helpers (same results as the reference's)
# ----------------------------------------------------------------------------------------------------------------------
def generate_background(W, H, mode="solid", color=(0.0, 0.0, 0.0), noise_range=(0, 255), seed=42):
    np.random.seed(seed)
    if mode == "solid":
        return np.full((H, W, 3), color, dtype=np.float32)
    lo, hi = noise_range
    if mode == "uniform_noise":
        return (np.random.uniform(lo, hi, (H, W, 3)) / 255.0).astype(np.float32)
    if mode == "gaussian_noise":
        noise = np.random.normal((lo + hi) / 2.0, (hi - lo) / 6.0, (H, W, 3))
        return (np.clip(noise, lo, hi) / 255.0).astype(np.float32)
    raise ValueError(f"Unknown background mode: {mode}")


def scale_intrinsics(*args):
    """(fx, fy, cx, cy, original_size, new_size) -> 4 values, or (f, cx, cy, original_size, new_size) -> 3 values;
    sizes are (H, W); x-like quantities scale with the width ratio, y-like ones with the height ratio."""
    if len(args) == 6:
        fx, fy, cx, cy, osz, nsz = args
        single = False
    elif len(args) == 5:
        fx, cx, cy, osz, nsz = args
        fy, single = fx, True
    else:
        raise ValueError("scale_intrinsics(fx, fy, cx, cy, original_size, new_size) or "
                         "scale_intrinsics(f, cx, cy, original_size, new_size)")
    oh, ow = osz
    nh, nw = nsz
    if oh == 0 or ow == 0:
        logger.warning("Original image size has a zero dimension, cannot scale intrinsics.")
        return (fx, cx, cy) if single else (fx, fy, cx, cy)
    sw, sh = nw / ow, nh / oh
    if single:
        return fx * sw, cx * sw, cy * sh
    return fx * sw, fy * sh, cx * sw, cy * sh


_ARIA_RADIAL = (-0.02340373583137989, 0.09388021379709244, -0.06088035926222801,
                0.0053304750472307205, 0.003342868760228157, -0.0006356257363222539)
_ARIA_TANGENTIAL = (0.0005087381578050554, -0.0004747129278257489)
_ARIA_THIN_PRISM = (-0.0011330085108056664, -0.00025734835071489215,
                    0.00009328465239377692, 0.00009424977179151028)


def load_aria_distortion_coeffs(online_calibration_path: Optional[str] = None, frame_idx: int = 0):
    """(radial[6], tangential[2], thin_prism[4], focal[1] or None, principal_point[2] or None).  Without a file: the
    preset Ego-Exo4D coefficients and (None, None).  With a file: record `frame_idx` (0 if out of range), camera
    'camera-rgb', projection 'FisheyeRadTanThinPrism' with params [f, cx, cy, k1..k6, p1, p2, s1..s4]; five Nones when
    the record does not have them."""
    if online_calibration_path is None:
        return (np.array(_ARIA_RADIAL), np.array(_ARIA_TANGENTIAL), np.array(_ARIA_THIN_PRISM), None, None)
    with open(online_calibration_path) as f:
        lines = f.readlines()
    none5 = (None, None, None, None, None)
    if not lines:
        logger.error("online_calibration.jsonl is empty.")
        return none5
    if frame_idx >= len(lines):
        frame_idx = 0
    rec = json.loads(lines[frame_idx])
    cam = next((c for c in rec.get("CameraCalibrations", []) if c.get("Label") == "camera-rgb"), None)
    if cam is None:
        logger.error("camera-rgb (Aria) calibration not found in online_calibration.jsonl")
        return none5
    proj = cam.get("Projection")
    if not proj or proj.get("Name") != "FisheyeRadTanThinPrism":
        return none5
    p = proj.get("Params", [])
    if len(p) < 10:
        return none5
    return (np.array(p[3:9]), np.array(p[9:11]), np.array(p[11:15]), np.array([p[0]]), np.array([p[1], p[2]]))


def load_camera_params_from_meta(meta_json_path: str, input_dir: str):
    """(K_exo (3,3) f32, exo W2C (4,4) f32, K_ego (3,3) f32, [ego W2C (4,4) f32 per frame]) of the sample whose
    exo_path's parent directory is named like Path(input_dir).stem; ValueError when there is none."""
    video_name = Path(input_dir).stem
    with open(meta_json_path) as f:
        meta = json.load(f)
    sample = next((s for s in meta.get("test_datasets", [])
                   if Path(s.get("exo_path", "")).parent.name == video_name), None)
    if sample is None:
        raise ValueError(f"No matching sample found for video_name='{video_name}' in {meta_json_path}")

    def to44(m):
        T = np.eye(4, dtype=np.float32)
        T[:3, :] = np.array(m, dtype=np.float32)
        return T

    return (np.array(sample["camera_intrinsics"], dtype=np.float32), to44(sample["camera_extrinsics"]),
            np.array(sample["ego_intrinsics"], dtype=np.float32), [to44(e) for e in sample["ego_extrinsics"]])


# ----------------------------------------------------------------------------------------------------------------------
# artifact walking
# ----------------------------------------------------------------------------------------------------------------------
def _first_artifact(input_dir) -> ArtifactPath:
    arts = list(ArtifactPath.glob_artifacts(Path(input_dir), use_video=True))
    if not arts:
        raise ValueError(f"No ViPE artifacts found in {input_dir}")
    return arts[0]


def _none_after_missing(it):
    """Like the reference's none_it: a missing zip turns the stream into an endless stream of (None, None)."""
    try:
        for item in it:
            yield item
    except FileNotFoundError:
        while True:
            yield None, None


class _RayTables:
    """Device ray tables of non-pinhole exo cameras, one per (camera type, intrinsics, frame size, subsample); None for
    PINHOLE, whose rays the kernel evaluates itself."""

    def __init__(self):
        self._cache = {}

    def get(self, camera_type, intr, H: int, W: int, s: int):
        ct = str(camera_type).upper()
        if ct == "PINHOLE":
            return None
        key = (ct, np.asarray(intr, dtype=np.float32).tobytes(), int(H), int(W), int(s))
        if key not in self._cache:
            import torch
            tab = ray_table(ct, intr, H, W, s)
            self._cache[key] = _engine().to_device(tab, torch.float32)
        return self._cache[key]


def _frames(artifact: ArtifactPath, tolerant: bool, wanted=None):
    """One sequential pass over (c2w, rgb, intr, camera type, depth, mask) like the reference's zip(...).  wanted: frame
    indices whose colour image, depth and mask are needed; for the others those three come back as io.SKIPPED,
    undecoded where the container allows it."""
    _, poses = read_pose_artifacts(artifact.pose_path)
    _, intr, types = read_intrinsics_artifacts(artifact.intrinsics_path, artifact.camera_type_path)
    depth_it = read_depth_artifacts(artifact.depth_path, keep_half=True, wanted=wanted)  # uploaded as stored
    mask_it = read_instance_artifacts(artifact.mask_path, wanted=wanted)
    if tolerant:
        depth_it, mask_it = _none_after_missing(depth_it), _none_after_missing(mask_it)
    for c2w, (_, rgb), k, ct, (_, depth), (_, mask) in zip(poses, read_rgb_artifacts(artifact.rgb_path, wanted=wanted),
                                                         intr, types, depth_it, mask_it):
        yield c2w, rgb, k, ct, depth, mask


# ----------------------------------------------------------------------------------------------------------------------
# cloud builders
# ----------------------------------------------------------------------------------------------------------------------
def _accumulated_background_clouds(frames, T_cam_to_world, spatial_subsample: int, want_xyz64: bool):
    """Device half of build_background_pointcloud: (one static cloud per frame that has depth, (H,W) or None)."""
    E = _engine()
    clouds, image_size = [], None
    tables, first = _RayTables(), None
    for c2w, rgb, intr, ct, depth, mask in frames:
        if depth is None:
            continue
        if image_size is None:
            image_size = tuple(rgb.shape[:2])
        if first is None:
            first = (intr, ct)  # the ray table is built once, from the first usable frame (reference :384)
        rays = tables.get(first[1], first[0], depth.shape[0], depth.shape[1], spatial_subsample)
        clouds.append(E.unproject_frame(depth, rgb, mask, first[0], c2w, T_cam_to_world, subsample=spatial_subsample,
                                        select=_lib.SELECT_STATIC, want_xyz64=want_xyz64, rays=rays))
    return clouds, image_size


def _concat_clouds(clouds):
    """One device cloud holding the live points of `clouds` in order (one host synchronisation for all the counts)."""
    import torch
    E = _engine()
    counts = torch.cat([c.count for c in clouds]).cpu().tolist()
    out = E.new_cloud(sum(counts))
    if sum(counts):
        torch.cat([c.points[:n] for c, n in zip(clouds, counts)], out=out.points)
    out.count.fill_(sum(counts))
    # "some colour byte exceeds 1" (the renderers' divide-by-255 rule, :812-815) holds for the whole when it holds for a part
    out.flags.copy_(torch.cat([c.flags for c, n in zip(clouds, counts) if n] or [out.flags]).max().reshape(1))
    return out


def build_background_pointcloud(input_dir: str, T_cam_to_world: np.ndarray, spatial_subsample: int = 2):
    """Static cloud accumulated over all frames: (points (N,3) float64, colours (N,3) uint8, (H,W), T_cam_to_world) --
    a 4-tuple on success like the reference (:449); 3-tuple with empty arrays when no frame has depth (:452-456)."""
    artifact = _first_artifact(input_dir)
    clouds, image_size = _accumulated_background_clouds(_frames(artifact, tolerant=True), T_cam_to_world,
                                                        spatial_subsample, want_xyz64=True)
    pts, cols = [], []
    for cloud in clouds:
        n = cloud.n()
        pts.append(cloud.xyz64[:n].cpu().numpy())
        cols.append(cloud.to_numpy()[1])
    if pts:
        return np.concatenate(pts, axis=0), np.concatenate(cols, axis=0), image_size, T_cam_to_world
    logger.warning("No valid background points found in any frame")
    return np.empty((0, 3)), np.empty((0, 3)), image_size


def compute_robust_mean_tensors(stacked_depth, stacked_rgb):
    """torch.nanmean over the frame axis with NaN -> 0: the two torch calls the reference makes (:466-471)."""
    import torch
    return (torch.nan_to_num(torch.nanmean(stacked_depth, dim=0), nan=0.0),
            torch.nan_to_num(torch.nanmean(stacked_rgb, dim=0), nan=0.0))


def build_mean_background_pointcloud(input_dir: str, T_cam_to_world: np.ndarray, spatial_subsample: int = 2,
                                     max_frames: int = None):
    """Static cloud from the per-pixel NaN-aware mean over frames: (points (N,3) float64, colours (N,3) uint8, (H,W)).
    Frames stream through a 17-plane accumulator in HBM instead of being stacked (the reference holds 16*F*H*W bytes)."""
    artifact = _first_artifact(input_dir)
    cloud, image_size = _mean_background_cloud(_frames(artifact, tolerant=True), T_cam_to_world, spatial_subsample,
                                               max_frames, want_xyz64=True)
    if cloud is None:
        logger.warning("No valid background frames found.")
        return np.empty((0, 3)), np.empty((0, 3)), image_size
    n = cloud.n()
    return cloud.xyz64[:n].cpu().numpy(), cloud.to_numpy()[1], image_size


def _mean_background_cloud(frames, T_cam_to_world, spatial_subsample: int, max_frames, want_xyz64: bool):
    """Device half of build_mean_background_pointcloud: (cloud or None when no frame has depth, (H,W) or None)."""
    E = _engine()
    state, image_size, n_used = None, None, 0
    first = None
    for c2w, rgb, intr, ct, depth, mask in frames:
        if first is None:
            first = (c2w, intr, ct)
        if depth is None:
            continue
        if max_frames and n_used >= max_frames:
            break
        if image_size is None:
            image_size = tuple(rgb.shape[:2])
            state = E.meanbg_new_state(*image_size)
        E.meanbg_accumulate(state, depth, rgb, mask, n_used)
        n_used += 1
    if state is None:
        return None, image_size
    c2w, intr, ct = first
    mean_d, mean_c = E.meanbg_finalize(state)
    rays = _RayTables().get(ct, intr, mean_d.shape[0], mean_d.shape[1], spatial_subsample)
    cloud = E.unproject_frame(mean_d, mean_c, None, intr, c2w, T_cam_to_world, subsample=spatial_subsample,
                              select=_lib.SELECT_ALL, want_xyz64=want_xyz64, rays=rays)
    return cloud, image_size


def build_dynamic_points_for_frame(artifact_path: ArtifactPath, frame_idx: int, T_cam_to_world: np.ndarray,
                                   spatial_subsample: int = 2):
    """Dynamic (instance != 0) points of one frame: (points (M,3) float64, colours (M,3) uint8); empty float arrays when
    the frame has no depth or no mask.  A missing depth/mask zip raises FileNotFoundError (reference :675)."""
    E = _engine()
    for idx, (c2w, rgb, intr, ct, depth, mask) in enumerate(_frames(artifact_path, tolerant=False)):
        if idx != frame_idx:
            continue
        if depth is None or mask is None:
            return np.empty((0, 3)), np.empty((0, 3))
        rays = _RayTables().get(ct, intr, depth.shape[0], depth.shape[1], spatial_subsample)
        cloud = E.unproject_frame(depth, rgb, mask, intr, c2w, T_cam_to_world, subsample=spatial_subsample,
                                  select=_lib.SELECT_DYNAMIC, want_xyz64=True, rays=rays)
        n = cloud.n()
        if n == 0:
            return np.empty((0, 3)), np.empty((0, 3))
        return cloud.xyz64[:n].cpu().numpy(), cloud.to_numpy()[1]
    return None  # the reference falls off the end of its loop the same way


# ----------------------------------------------------------------------------------------------------------------------
# renderers
# ----------------------------------------------------------------------------------------------------------------------
def _colors_u8(colors_world, n: int) -> np.ndarray:
    c = np.asarray(colors_world)
    if c.size == 0:
        return np.zeros((0, 3), dtype=np.uint8)
    c = c.reshape(-1, 3)
    if c.shape[0] != n:
        raise ValueError("points and colours differ in length")
    if c.dtype != np.uint8:
        cf = np.asarray(c, dtype=np.float64)
        if not (np.all(cf == np.round(cf)) and cf.min() >= 0 and cf.max() <= 255):
            raise ValueError("colours must be integers in 0..255")
        c = cf.astype(np.uint8)
    return c


def _upload(points_world, colors_world):
    pts = np.asarray(points_world, dtype=np.float32).reshape(-1, 3)  # the reference's float32 cast (:800, :953)
    return _engine().filter_cloud(pts, _colors_u8(colors_world, pts.shape[0]))


def _background_u8(W, H, mode, color, noise_range, seed):
    bg = generate_background(W, H, mode, color, noise_range, seed) if mode != "solid" \
        else np.full((H, W, 3), color, dtype=np.float32)
    return bg


def _is_black(background_mode, background_color) -> bool:
    return background_mode == "solid" and tuple(float(c) for c in background_color) == (0.0, 0.0, 0.0)


def _finish(res, rotate_cw, W, H, background_mode, background_color, noise_range, seed) -> np.ndarray:
    """The uint8 frame of a render call: the kernel's own for the black background the reference's main always passes,
    otherwise the compositor's float32 output post-processed against the generated background image (:794-797, :873-888)."""
    if _is_black(background_mode, background_color):
        return res["image"].cpu().numpy()
    bg = _background_u8(W, H, background_mode, background_color, noise_range, seed)
    return _engine().postprocess_background(res["image_f32"], bg, rotate_cw).cpu().numpy()


def render_points_pytorch3d(points_world, colors_world, K, T_c2w=None, T_w2c=None, W=640, H=480, point_size=2,
                            device="cuda", background_mode="solid", background_color=(0.0, 0.0, 0.0),
                            noise_range=(0, 255), seed=42, _cloud=None, _dyn=None):
    """One pinhole ego frame, uint8, rotated 90 degrees clockwise ((W,H,3)); (H,W,3) background when no point
    survives the finite / |p| < 1e6 filters."""
    cam, rp = _pinhole_setup(K, T_c2w, T_w2c, W, H, point_size)
    E = _engine()
    cloud = _cloud if _cloud is not None else _upload(points_world, colors_world)
    if cloud.n() + (_dyn.n() if _dyn is not None else 0) == 0:
        return (_background_u8(W, H, background_mode, background_color, noise_range, seed) * 255).astype(np.uint8)
    black = _is_black(background_mode, background_color)
    res = E.render_frame(cam, rp, cloud, _dyn, want_f32=not black)
    return _finish(res, True, W, H, background_mode, background_color, noise_range, seed)


def _pinhole_setup(K, T_c2w, T_w2c, W, H, point_size):
    """Argument checks and camera / raster parameters of render_points_pytorch3d (same assertions, same order)."""
    if T_w2c is None:
        assert T_c2w is not None, "Provide either T_w2c or T_c2w."
        T_w2c = np.linalg.inv(T_c2w)
    T_w2c = np.asarray(T_w2c)
    R_cv, t_cv = _eng.split_w2c(T_w2c)
    assert np.isfinite(R_cv).all(), "R has NaN/Inf"
    assert np.isfinite(t_cv).all(), "t has NaN/Inf"
    assert np.isfinite(np.asarray(K)).all(), "K has NaN/Inf"
    return _eng.pinhole_camera(R_cv, t_cv, K, H, W), _eng.make_raster_params(point_size, H, W, K=20, rotate_cw=True)


def render_points_fisheye(points_world, colors_world, T_w2c, ego_intrinsics, W=640, H=480, point_size=2,
                          device="cuda", radial_distortion_coeffs=None, tangential_distortion_coeffs=None,
                          thinPrism_distortion_coeffs=None, focal_length=None, principal_point=None,
                          original_image_size=None, background_mode="solid", background_color=(0.0, 0.0, 0.0),
                          noise_range=(0, 255), seed=42, _cloud=None, _dyn=None):
    """One fisheye (FisheyeRadTanThinPrism) ego frame, uint8 (H,W,3), not rotated."""
    T_w2c = np.asarray(T_w2c)
    R_cv, t_cv = _eng.split_w2c(T_w2c)
    assert np.isfinite(R_cv).all(), "R has NaN/Inf"
    assert np.isfinite(t_cv).all(), "t has NaN/Inf"
    E = _engine()
    cloud = _cloud if _cloud is not None else _upload(points_world, colors_world)
    if cloud.n() + (_dyn.n() if _dyn is not None else 0) == 0:
        return (_background_u8(W, H, background_mode, background_color, noise_range, seed) * 255).astype(np.uint8)
    cam, rp = _fisheye_setup(R_cv, t_cv, ego_intrinsics, W, H, point_size, radial_distortion_coeffs,
                             tangential_distortion_coeffs, thinPrism_distortion_coeffs, focal_length, principal_point,
                             original_image_size)
    black = _is_black(background_mode, background_color)
    res = E.render_frame(cam, rp, cloud, _dyn, want_f32=not black)
    return _finish(res, False, W, H, background_mode, background_color, noise_range, seed)


def _fisheye_setup(R_cv, t_cv, ego_intrinsics, W, H, point_size, radial_distortion_coeffs, tangential_distortion_coeffs,
                   thinPrism_distortion_coeffs, focal_length, principal_point, original_image_size):
    """Camera / raster parameters of render_points_fisheye (same checks)."""
    for name, c in (("radial_distortion_coeffs", radial_distortion_coeffs),
                    ("tangential_distortion_coeffs", tangential_distortion_coeffs),
                    ("thinPrism_distortion_coeffs", thinPrism_distortion_coeffs)):
        if c is None or len(c) == 0:
            raise ValueError(f"{name} must be provided")
    if focal_length is not None and principal_point is not None:
        f, cx, cy = focal_length[0], principal_point[0], principal_point[1]
        if original_image_size:
            f, cx, cy = scale_intrinsics(f, cx, cy, original_image_size, (H, W))
    else:
        f, cx, cy = ego_intrinsics[0, 0], ego_intrinsics[0, 2], ego_intrinsics[1, 2]
    cam = _eng.fisheye_camera(R_cv, t_cv, f, cx, cy, radial_distortion_coeffs, tangential_distortion_coeffs,
                              thinPrism_distortion_coeffs, H, W)
    return cam, _eng.make_raster_params(point_size, H, W, K=20, rotate_cw=False)


def project_points_to_image_sequential(bg_points_3d: np.ndarray, bg_colors: np.ndarray,
                                       ego_extrinsics_list: List[np.ndarray], ego_intrinsics: np.ndarray,
                                       image_size: Tuple[int, int], point_size: float = 1.0,
                                       use_fisheye: bool = False, online_calibration_path: str = None,
                                       original_image_size: Optional[Tuple[int, int]] = None,
                                       artifact_path: Optional[ArtifactPath] = None,
                                       T_cam_to_world: Optional[np.ndarray] = None,
                                       only_bg: bool = False) -> List[np.ndarray]:
    """The frame loop: per ego pose, [static cloud | that frame's dynamic points] rendered pinhole or fisheye."""
    height, width = image_size
    num_frames = len(ego_extrinsics_list)
    if len(bg_points_3d) == 0 and artifact_path is None:
        return [np.zeros((height, width, 3), dtype=np.uint8) for _ in range(num_frames)]
    coeffs, aria_size = _lens_setup(use_fisheye, online_calibration_path, original_image_size)
    bg = _upload(bg_points_3d, bg_colors)  # uploaded and filtered once; the filters are per point
    want_dyn = (not only_bg) and artifact_path is not None and T_cam_to_world is not None
    frames = _frames(artifact_path, tolerant=False) if want_dyn else None
    return _render_sequence(bg, frames, ego_extrinsics_list, ego_intrinsics, image_size, point_size, use_fisheye,
                            coeffs, aria_size, T_cam_to_world)


def _lens_setup(use_fisheye, online_calibration_path, original_image_size):
    """((radial, tangential, thin prism, focal, principal point), Aria sensor size) of the frame loop (:1142-1160)."""
    aria_size = original_image_size if original_image_size is not None else (2880, 2880)
    rad = tan = thin = foc = pp = None
    if use_fisheye:
        rad, tan, thin, foc, pp = load_aria_distortion_coeffs(online_calibration_path, frame_idx=0)
        if rad is None or tan is None or thin is None:
            raise ValueError("Failed to load distortion coefficients.")
    return (rad, tan, thin, foc, pp), aria_size


def _render_sequence(bg, frames, ego_extrinsics_list, ego_intrinsics, image_size, point_size, use_fisheye, coeffs,
                     aria_size, T_cam_to_world) -> List[np.ndarray]:
    """The frame loop proper on a static cloud that is already in HBM.  `frames`: None (static cloud only) or an iterator
    of (c2w, rgb, intr, camera type, depth, mask), one item per ego pose, whose dynamic points join the static cloud."""
    height, width = image_size
    num_frames = len(ego_extrinsics_list)
    rad, tan, thin, foc, pp = coeffs
    E = _engine()
    want_dyn = frames is not None
    tables = _RayTables()

    def camera(T_w2c):
        if use_fisheye:
            T = np.asarray(T_w2c)
            R_cv, t_cv = _eng.split_w2c(T)
            assert np.isfinite(R_cv).all(), "R has NaN/Inf"
            assert np.isfinite(t_cv).all(), "t has NaN/Inf"
            return _fisheye_setup(R_cv, t_cv, ego_intrinsics, width, height, point_size, rad, tan, thin, foc, pp,
                                  aria_size)
        return _pinhole_setup(ego_intrinsics, None, T_w2c, width, height, point_size)

    if bg.n() == 0:
        # Without static points a frame may end up empty, which the reference answers with an unrotated background
        # image: that needs each frame's point count on the host, so these frames are rendered one at a time.
        out = []
        for frame_idx, T_w2c in enumerate(ego_extrinsics_list):
            dyn = None
            item = next(frames, None) if want_dyn else None
            if item is not None:
                c2w, rgb, intr, ct, depth, mask = item
                if depth is not None and mask is not None:
                    dyn = E.unproject_frame(depth, rgb, mask, intr, c2w, T_cam_to_world, subsample=2,
                                            select=_lib.SELECT_DYNAMIC,
                                            rays=tables.get(ct, intr, depth.shape[0], depth.shape[1], 2))
            if use_fisheye:
                img = render_points_fisheye(None, None, T_w2c, ego_intrinsics, W=width, H=height,
                                            point_size=point_size, radial_distortion_coeffs=rad,
                                            tangential_distortion_coeffs=tan, thinPrism_distortion_coeffs=thin,
                                            focal_length=foc, principal_point=pp, original_image_size=aria_size,
                                            _cloud=bg, _dyn=dyn)
            else:
                img = render_points_pytorch3d(None, None, K=ego_intrinsics, T_w2c=T_w2c, W=width, H=height,
                                              point_size=point_size, _cloud=bg, _dyn=dyn)
            out.append(img)
        return out

    # The static cloud is not empty, so no frame is: frames go through the pipeline with several of them in flight and
    # come back to the host together at the end.
    import torch
    pipe = None
    images = None
    keep = []  # uploaded exo frames stay alive until the pipeline has finished with them
    for frame_idx, T_w2c in enumerate(ego_extrinsics_list):
        cam, rp = camera(T_w2c)
        item = next(frames, None) if want_dyn else None
        dyn_in = None
        if item is not None:
            c2w, rgb, intr, ct, depth, mask = item
            if depth is not None and mask is not None:
                d = E.depth_to_device(depth)
                dyn_in = (d, E.to_device(rgb, torch.uint8), E.to_device(mask, torch.uint8),
                          _eng.make_unproject_params(d.shape[0], d.shape[1], 2, _lib.SELECT_DYNAMIC, intr, c2w,
                                                     T_cam_to_world),
                          tables.get(ct, intr, d.shape[0], d.shape[1], 2))
        if pipe is None:
            pipe = _eng.FramePipeline(E, bg, rp, height, width,
                                      exo_hw=tuple(dyn_in[0].shape) if dyn_in is not None else None, subsample=2)
            images = torch.empty((num_frames,) + tuple(pipe.image_shape), dtype=torch.uint8, device=E.device)
        if dyn_in is not None:
            if not (isinstance(depth, torch.Tensor) and depth.is_cuda):  # frames of a resident clip outlive the loop
                keep.append(dyn_in)
            pipe.submit(cam, images[frame_idx], up=dyn_in[3], depth=dyn_in[0], rgb=dyn_in[1], mask=dyn_in[2],
                        after_main=True, rays=dyn_in[4])
        else:
            pipe.submit(cam, images[frame_idx])
        if len(keep) >= 4 * pipe.n_slots:  # bound the device memory held by uploaded frames
            pipe.join()
            torch.cuda.current_stream(E.device).synchronize()
            keep.clear()
    if pipe is None:
        return []
    pipe.finish()
    host = images.cpu().numpy()
    return [host[i] for i in range(num_frames)]


# ----------------------------------------------------------------------------------------------------------------------
# CLI
# ----------------------------------------------------------------------------------------------------------------------
def get_parser():
    p = argparse.ArgumentParser(description="Render ViPE point cloud from ego view")
    p.add_argument("--input_dir", required=True, help="Directory containing ViPE artifacts")
    p.add_argument("--out_dir", required=True, help="Output directory for rendered images")
    p.add_argument("--meta_json_path", required=True, help="Path to meta.json file containing camera parameters")
    p.add_argument("--point_size", type=float, default=1.0, help="Size of rendered points")
    p.add_argument("--start_frame", type=int, default=0, help="Starting frame number for rendering (default: 0)")
    p.add_argument("--end_frame", type=int, required=True, help="Ending frame number for rendering (inclusive)")
    p.add_argument("--batch_size", type=int, default=8, help="Batch size for rendering (default: 8)")
    p.add_argument("--only_bg", action="store_true", help="Only render background points")
    p.add_argument("--use_mean_bg", action="store_true", help="Use nanmean background instead of standard background")
    p.add_argument("--fish_eye_rendering", action="store_true", help="Enable fish-eye rendering")
    p.add_argument("--online_calibration_path", type=str, default=None,
                   help="(Optional) online_calibration.jsonl with the real Aria distortion coefficients")
    # extras of this implementation, all opt-in
    p.add_argument("--dump_frames", action="store_true",
                   help="also write frame_%%06d.png next to the video and print the white-pixel metrics the batch "
                        "scripts derive from such a dump (data_preprocess/scripts/postprocess_render_dir.py)")
    p.add_argument("--white_threshold", type=float, default=30.0, help="grey threshold of the white-pixel metrics")
    p.add_argument("--dist_backend", choices=["nccl", "gloo"], default="nccl",
                   help="under torchrun only: nccl (RCCL over xGMI, one rank per GPU) or gloo, with which several ranks "
                        "may share a GPU (a rehearsal of the multi-GPU path on a one-GPU machine)")
    return p


def configure_output_directory(args) -> str:
    try:
        return os.path.join(args.out_dir, Path(args.input_dir).stem)
    except Exception:
        return args.out_dir


def get_inference_frame_range(input_dir: str):
    artifact = _first_artifact(input_dir)
    if not artifact.pose_path.exists():
        raise FileNotFoundError(f"Pose file not found: {artifact.pose_path}")
    inds, _ = read_pose_artifacts(artifact.pose_path)
    if len(inds) == 0:
        raise ValueError("No pose data found in inference results")
    return int(inds.min()), int(inds.max())


def validate_frame_range(args, total_frames_available: int, input_dir: str):
    lo, hi = get_inference_frame_range(input_dir)
    if args.start_frame < 0:
        raise ValueError(f"start_frame must be non-negative, got {args.start_frame}")
    if args.end_frame <= args.start_frame:
        raise ValueError(f"end_frame ({args.end_frame}) must be greater than start_frame ({args.start_frame})")
    if args.start_frame != lo:
        raise ValueError(f"start_frame ({args.start_frame}) must match inference range start ({lo})")
    if args.end_frame != hi:
        raise ValueError(f"end_frame ({args.end_frame}) must match inference range end ({hi})")
    if args.start_frame >= total_frames_available:
        raise ValueError(f"start_frame ({args.start_frame}) exceeds available frames ({total_frames_available})")
    if args.end_frame >= total_frames_available:
        raise ValueError(f"end_frame ({args.end_frame}) exceeds available frames ({total_frames_available})")


def main(argv=None):
    logging.basicConfig(level=logging.INFO)
    args = get_parser().parse_args(argv)
    args.out_dir = configure_output_directory(args)
    os.makedirs(args.out_dir, exist_ok=True)
    _, exo_extrinsic, ego_intrinsic, ego_list = load_camera_params_from_meta(args.meta_json_path, args.input_dir)
    T_cam_to_world = np.linalg.inv(exo_extrinsic)
    fixed_image_size = (448, 448)
    from .clip import DeviceClip
    artifact = _first_artifact(args.input_dir)
    clip = None if os.environ.get("EGOPRIOR_STREAM_CLIP") == "1" else DeviceClip.load(_engine(), artifact)
    if clip is not None:
        # The clip is decoded once into HBM; the background cloud is built from it and stays on the GPU, and the frame
        # loop takes its exo frames from the same tensors.  Same kernels on the same values as the two public functions
        # below produce through their numpy returns.
        if args.use_mean_bg:
            bg, _ = _mean_background_cloud(clip.frames(tolerant=True), T_cam_to_world, 2, None, want_xyz64=False)
        else:
            clouds, _ = _accumulated_background_clouds(clip.frames(tolerant=True), T_cam_to_world, 2, want_xyz64=False)
            if not clouds:
                logger.warning("No valid background points found in any frame")
            bg = _concat_clouds(clouds) if clouds else None
        if bg is None or bg.n() == 0:
            logger.error("No points in background point cloud. Exiting.")
            return
        validate_frame_range(args, len(ego_list), args.input_dir)
        coeffs, aria_size = _lens_setup(args.fish_eye_rendering, args.online_calibration_path, (2880, 2880))
        frames = _render_sequence(bg, None if args.only_bg else clip.frames(tolerant=False),
                                  ego_list[args.start_frame:args.end_frame + 1], ego_intrinsic, fixed_image_size,
                                  args.point_size, args.fish_eye_rendering, coeffs, aria_size, T_cam_to_world)
    else:
        if args.use_mean_bg:
            res = build_mean_background_pointcloud(args.input_dir, T_cam_to_world)
        else:
            res = build_background_pointcloud(args.input_dir, T_cam_to_world)
        bg_pts, bg_cols = res[0], res[1]
        if len(bg_pts) == 0:
            logger.error("No points in background point cloud. Exiting.")
            return
        validate_frame_range(args, len(ego_list), args.input_dir)
        frames = project_points_to_image_sequential(
            bg_pts, bg_cols, ego_list[args.start_frame:args.end_frame + 1], ego_intrinsic, fixed_image_size,
            args.point_size, use_fisheye=args.fish_eye_rendering, online_calibration_path=args.online_calibration_path,
            original_image_size=(2880, 2880), artifact_path=artifact, T_cam_to_world=T_cam_to_world,
            only_bg=args.only_bg)
    out = write_video(Path(args.out_dir) / "ego_Prior.mp4", frames, fps=30)
    logger.info(f"Rendering complete. Saved {len(frames)} frames to {out}")
    if args.dump_frames and frames:
        dump_frames_and_metrics(frames, args.out_dir, args.white_threshold)


def dump_frames_and_metrics(frames, out_dir, threshold: float = 30.0, write_masks: bool = True):
    """What infer_vipe_all_takes.sh:351-356 (ffmpeg frame dump) followed by postprocess_render_dir.py:34-67 produce:
    frame_%06d.png (ffmpeg numbers from 1), frame_%06d_mask.png and the three metric lines.  The grey conversion and the
    counts run on the GPU; returns (total_white_pixels, frames_with_any_white, num_frames)."""
    from .io import encode_png_gray, encode_png_rgb
    E = _engine()
    stack = np.stack([np.asarray(f, dtype=np.uint8) for f in frames])
    counts, masks = E.white_pixel_counts(stack, threshold, want_masks=True)
    counts, masks = counts.cpu().numpy(), masks.cpu().numpy()
    out_dir = Path(out_dir)
    for i, f in enumerate(frames):
        (out_dir / f"frame_{i + 1:06d}.png").write_bytes(encode_png_rgb(np.asarray(f, dtype=np.uint8)))
        if write_masks:
            (out_dir / f"frame_{i + 1:06d}_mask.png").write_bytes(encode_png_gray(masks[i]))
    total, any_white = int(counts.astype(np.int64).sum()), int((counts > 0).sum())
    print(f"total_white_pixels={total}")
    print(f"frames_with_any_white={any_white}")
    print(f"num_frames={len(frames)}")
    return total, any_white, len(frames)


if __name__ == "__main__":
    main()
