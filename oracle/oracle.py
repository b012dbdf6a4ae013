"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A numpy (+ small C library, oracle/egoprior_oracle.c) restatement of the arithmetic on the reference's
ego-prior rendering path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  The product path (the egoprior package + its HIP library) never
imports it and has no CPU fallback.

What each function follows (paths relative to the reference checkout):
  pinhole_rays ................ vipe/utils/cameras.py:131-159, scripts/render_vipe_pointcloud.py:384-398
  reliable_depth_mask_range ... vipe/utils/depth.py:385-422
  unproject_frame ............. scripts/render_vipe_pointcloud.py:358-445 (static) and :682-751 (dynamic)
  nanmean_stack / mean_background / build_mean_background_points
                                scripts/render_vipe_pointcloud.py:459-473, 522-656
  filter_points ............... scripts/render_vipe_pointcloud.py:800-834, 953-979
  project_pinhole ............. scripts/render_vipe_pointcloud.py:840-850 + pytorch3d cameras_from_opencv_projection
                                / PerspectiveCameras / PointsRasterizer.transform (restated, v0.7.9)
  project_fisheye ............. scripts/render_vipe_pointcloud.py:990-1062 + pytorch3d FishEyeCameras (restated)
  radius_ndc .................. scripts/render_vipe_pointcloud.py:853-855, 1065-1067
  rasterize / composite ....... pytorch3d rasterize_points (naive CPU) + PointsRenderer + AlphaCompositor (restated,
                                see egoprior_oracle.c)
  postprocess ................. scripts/render_vipe_pointcloud.py:874-888, 1086-1102

Pinning status:
  * pinhole_rays, reliable_depth_mask_range, nanmean_stack: PINNED against outputs of the reference's own
    vipe/utils/cameras.py, vipe/utils/depth.py and of torch.nanmean, generated in the dev container by
    tests/golden/make_golden.py and committed under tests/golden/.
  * everything that restates pytorch3d (project_*, rasterize, composite): PARITY UNPINNED -- the reference holds no
    tests or fixtures for that boundary and pytorch3d is not available; checked by hand-derived known answers only.

All arithmetic is float32 unless the reference does it in float64 (the exo->world transform), with a fixed,
written-down operation order and no fused multiply-add, so the HIP kernels can match it bit for bit.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

F32 = np.float32
_HERE = Path(__file__).resolve().parent
_LIB = None


def build_lib(force: bool = False) -> Path:
    so = _HERE / "libegoprior_oracle.so"
    src = _HERE / "egoprior_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-s", "-B", "libegoprior_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = build_lib()
        L = ctypes.CDLL(str(so))
        i32, i64, f32 = ctypes.c_int, ctypes.c_int64, ctypes.c_float
        vp = ctypes.c_void_p
        L.oracle_pix_to_ndc.restype = f32
        L.oracle_pix_to_ndc.argtypes = [i32, i32, i32]
        L.oracle_rasterize_points_naive.restype = None
        L.oracle_rasterize_points_naive.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp]
        L.oracle_rasterize_points_scatter.restype = None
        L.oracle_rasterize_points_scatter.argtypes = [vp, f32, i64, i32, i32, i32, vp, vp, vp]
        L.oracle_alpha_composite.restype = None
        L.oracle_alpha_composite.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp]
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------------------------------------
# R4: ray table
# --------------------------------------------------------------------------------------------------------------------
def pinhole_rays(H: int, W: int, s: int, intr) -> np.ndarray:
    """(ceil(H/s), ceil(W/s), 3) float32 rays X=(u-cx)/fx, Y=(v-cy)/fy, 1 at integer pixel coordinates."""
    fx, fy, cx, cy = [F32(v) for v in np.asarray(intr, dtype=F32)[:4]]
    u = np.arange(W, dtype=F32)[::s]
    v = np.arange(H, dtype=F32)[::s]
    X = (u - cx) / fx
    Y = (v - cy) / fy
    rays = np.empty((v.size, u.size, 3), dtype=F32)
    rays[..., 0] = X[None, :]
    rays[..., 1] = Y[:, None]
    rays[..., 2] = F32(1.0)
    # the reference divides by rays[..., 2:3] (= 1) for non-panorama cameras: a no-op
    return rays


def mei_rays(H: int, W: int, s: int, intr5) -> np.ndarray:
    """Unified (Mei) camera model, vipe/utils/cameras.py:228-250, on the pixel grid of
    scripts/render_vipe_pointcloud.py:384-398, followed by the script's division by z (= 1): float32 throughout.
        u_ = (u - cx) / fx, v_ = (v - cy) / fy, r2 = u_^2 + v_^2, q = sqrt(1 + (1 - k1^2) r2),
        factor = (k1 + q) / (1 + r2), X = u_ * factor / (factor - k1), Y likewise, Z = 1."""
    fx, fy, cx, cy, k1 = [F32(v) for v in np.asarray(intr5, dtype=F32)[:5]]
    v, u = np.meshgrid(np.arange(H, dtype=F32)[::s], np.arange(W, dtype=F32)[::s], indexing="ij")
    ub = (u - cx) / fx
    vb = (v - cy) / fy
    r2 = ub * ub + vb * vb
    q = np.sqrt(F32(1) + (F32(1) - k1 * k1) * r2)
    factor = (k1 + q) / (F32(1) + r2)
    X = ub * factor / (factor - k1)
    Y = vb * factor / (factor - k1)
    one = np.ones_like(X)
    rays = np.stack([X, Y, one], axis=-1).astype(F32)
    return rays / rays[..., 2:3]


def panorama_rays(H: int, W: int, s: int) -> np.ndarray:
    """Equirectangular model, vipe/utils/cameras.py:366-387, with u, v normalised by (W-1), (H-1) as at
    scripts/render_vipe_pointcloud.py:391-393; not divided by z.  sin / cos come from the host's libm here and from
    torch's vectorised kernels in the reference: equal to within an ulp or two, not always bit for bit."""
    v, u = np.meshgrid(np.arange(H, dtype=F32)[::s], np.arange(W, dtype=F32)[::s], indexing="ij")
    v = v / F32(H - 1)
    u = u / F32(W - 1)
    theta = (u - F32(0.5)) * F32(2) * F32(np.pi)
    phi = v * F32(np.pi)
    sp = np.sin(phi).astype(F32)
    x = sp * np.sin(theta).astype(F32)
    y = -np.cos(phi).astype(F32)
    z = sp * np.cos(theta).astype(F32)
    return np.stack([x, y, z], axis=-1).astype(F32)


# --------------------------------------------------------------------------------------------------------------------
# R5: reliability mask
# --------------------------------------------------------------------------------------------------------------------
def reliable_depth_mask_range(depth: np.ndarray, window: int = 5, ratio_thresh: float = 0.1, eps: float = 1e-6):
    """5x5 (max-min)/(mean+eps) < 0.1 and depth > 0.

    max/min pools ignore the border (-inf/+inf padding) and propagate NaN; the mean pool pads with zeros and always
    divides by window**2; its 25 taps are added one by one, rows first, in float32.
    """
    d = np.ascontiguousarray(depth, dtype=F32)
    H, W = d.shape
    p = window // 2
    pad_lo = np.full((H + 2 * p, W + 2 * p), -np.inf, dtype=F32)
    pad_hi = np.full((H + 2 * p, W + 2 * p), np.inf, dtype=F32)
    pad_0 = np.zeros((H + 2 * p, W + 2 * p), dtype=F32)
    pad_lo[p:p + H, p:p + W] = d
    pad_hi[p:p + H, p:p + W] = d
    pad_0[p:p + H, p:p + W] = d
    mx = np.full((H, W), -np.inf, dtype=F32)
    mn = np.full((H, W), np.inf, dtype=F32)
    acc = np.zeros((H, W), dtype=F32)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        for kh in range(window):
            for kw in range(window):
                mx = np.maximum(mx, pad_lo[kh:kh + H, kw:kw + W])  # np.maximum/minimum propagate NaN
                mn = np.minimum(mn, pad_hi[kh:kh + H, kw:kw + W])
                acc = acc + pad_0[kh:kh + H, kw:kw + W]
        mean = acc / F32(window * window)
        ratio = (mx - mn) / (mean + F32(eps))
        return (ratio < F32(ratio_thresh)) & (d > 0)


# --------------------------------------------------------------------------------------------------------------------
# R6 / R7: per-frame unprojection
# --------------------------------------------------------------------------------------------------------------------
def apply_c2w_f32(p_cam: np.ndarray, c2w: np.ndarray) -> np.ndarray:
    """R @ p + t in float32 with the fixed order ((r0*x + r1*y) + r2*z) + t."""
    c = np.asarray(c2w, dtype=F32)
    x, y, z = p_cam[..., 0], p_cam[..., 1], p_cam[..., 2]
    out = np.empty_like(p_cam, dtype=F32)
    for i in range(3):
        out[..., i] = ((c[i, 0] * x + c[i, 1] * y) + c[i, 2] * z) + c[i, 3]
    return out


def apply_T_f64(p: np.ndarray, T: np.ndarray) -> np.ndarray:
    """[p,1] @ T.T in float64 with the fixed order ((t0*x + t1*y) + t2*z) + t3."""
    Td = np.asarray(T, dtype=np.float64)
    pd = np.asarray(p, dtype=np.float64)
    x, y, z = pd[..., 0], pd[..., 1], pd[..., 2]
    out = np.empty(pd.shape, dtype=np.float64)
    for i in range(3):
        out[..., i] = ((Td[i, 0] * x + Td[i, 1] * y) + Td[i, 2] * z) + Td[i, 3]
    return out


def unproject_frame(depth, rgb_u8, mask, intr, c2w, T_cam_to_world, s: int = 2, select: str = "static", rays=None):
    """One frame's points: (N,3) float64 world coordinates, (N,3) uint8 colours, row-major over the [::s, ::s] grid.

    select="static" keeps mask == 0 (everything when mask is None); "dynamic" keeps mask != 0 (nothing when None).
    rays: ray table of a non-pinhole exo camera (mei_rays / panorama_rays); the pinhole table from `intr` otherwise.
    """
    depth = np.asarray(depth, dtype=F32)
    H, W = depth.shape
    if select == "dynamic" and mask is None:
        return np.empty((0, 3)), np.empty((0, 3))
    rays = pinhole_rays(H, W, s, intr) if rays is None else np.asarray(rays, dtype=F32)
    with np.errstate(invalid="ignore", over="ignore"):
        p_cam = rays * depth[::s, ::s, None]
        p_vipe = apply_c2w_f32(p_cam, c2w)
    valid = reliable_depth_mask_range(depth)[::s, ::s]
    if mask is not None:
        m = np.asarray(mask)[::s, ::s]
        valid = valid & ((m == 0) if select == "static" else (m != 0))
    cols = np.asarray(rgb_u8, dtype=np.uint8)[::s, ::s].reshape(-1, 3)[valid.reshape(-1)]
    pts = p_vipe.reshape(-1, 3)[valid.reshape(-1)]
    return apply_T_f64(pts, T_cam_to_world), cols


# --------------------------------------------------------------------------------------------------------------------
# R8: mean background
# --------------------------------------------------------------------------------------------------------------------
def _ceil_log2(x: int) -> int:
    if x <= 2:
        return 1
    return int(x - 1).bit_length()


def cascade_nansum(stack: np.ndarray) -> np.ndarray:
    """Sum over axis 0 ignoring NaN, in float32, in the order ATen's CPU cascade sum uses for an outer reduction:
    runs of 16 terms are added one by one into a level-0 accumulator, which is flushed into level 1 after each run,
    level 1 into level 2 every 16 runs, and so on (4 levels); the tail is added one by one, then levels are folded."""
    x = np.asarray(stack, dtype=F32)
    n = x.shape[0]
    num_levels = 4
    level_power = max(4, _ceil_log2(n) // num_levels)
    level_step = 1 << level_power
    level_mask = level_step - 1
    acc = [np.zeros(x.shape[1:], dtype=F32) for _ in range(num_levels)]

    def load(i):
        v = x[i]
        return np.where(np.isnan(v), F32(0), v)

    i = 0
    while i + level_step <= n:
        for _ in range(level_step):
            acc[0] = acc[0] + load(i)
            i += 1
        for j in range(1, num_levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = np.zeros_like(acc[0])
            if (i & (level_mask << (j * level_power))) != 0:
                break
    while i < n:
        acc[0] = acc[0] + load(i)
        i += 1
    for j in range(1, num_levels):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def nanmean_stack(stack: np.ndarray) -> np.ndarray:
    """torch.nan_to_num(torch.nanmean(stack, dim=0), nan=0) on CPU float32."""
    x = np.asarray(stack, dtype=F32)
    cnt = (~np.isnan(x)).sum(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = cascade_nansum(x) / cnt.astype(F32)
    return np.where(np.isnan(mean), F32(0), mean).astype(F32)


def mean_background(depths, rgbs_u8, masks):
    """(mean_depth (H,W) f32, mean_rgb (H,W,3) f32) over frames, NaN where (mask != 0) or depth unreliable."""
    depths = np.asarray(depths, dtype=F32)
    Fn = depths.shape[0]
    md = np.empty_like(depths)
    mr = np.empty(depths.shape + (3,), dtype=F32)
    for f in range(Fn):
        ok = reliable_depth_mask_range(depths[f])
        if masks is not None:
            ok = ok & (np.asarray(masks[f]) == 0)
        md[f] = np.where(ok, depths[f], F32(np.nan))
        rgb_f = np.asarray(rgbs_u8[f], dtype=F32) / F32(255.0)
        mr[f] = np.where(ok[..., None], rgb_f, F32(np.nan))
    return nanmean_stack(md), nanmean_stack(mr)


def build_mean_background_points(depths, rgbs_u8, masks, intr, c2w, T_cam_to_world, s: int = 2):
    """Static cloud from the per-pixel NaN-aware mean: (N,3) float64 points, (N,3) uint8 colours."""
    mean_depth, mean_rgb = mean_background(depths, rgbs_u8, masks)
    H, W = mean_depth.shape
    rays = pinhole_rays(H, W, s, intr)
    p_cam = rays * mean_depth[::s, ::s, None]
    p_vipe = apply_c2w_f32(p_cam, c2w)
    valid = reliable_depth_mask_range(mean_depth)[::s, ::s].reshape(-1)
    q = np.clip(mean_rgb * F32(255.0), F32(0), F32(255)).astype(np.uint8)  # truncation
    cols = q[::s, ::s].reshape(-1, 3)[valid]
    pts = p_vipe.reshape(-1, 3)[valid]
    return apply_T_f64(pts, T_cam_to_world), cols


# --------------------------------------------------------------------------------------------------------------------
# R9 / R9f: render-entry filters and cameras
# --------------------------------------------------------------------------------------------------------------------
def filter_points(points_world, colors_world):
    """float32 cast, drop non-finite rows, drop |p| >= 1e6, colours /255 when any colour value exceeds 1."""
    pts = np.asarray(points_world, dtype=F32).reshape(-1, 3)
    cols = np.asarray(colors_world, dtype=F32).reshape(-1, 3)
    finite = np.isfinite(pts).all(axis=1)
    pts, cols = pts[finite], cols[finite]
    with np.errstate(over="ignore"):
        mag = np.sqrt((pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2])
    keep = mag < F32(1e6)
    pts, cols = pts[keep], cols[keep]
    if cols.size and cols.max() > 1.0:
        cols = cols / F32(255.0)
    return pts, cols


def split_w2c(T_w2c):
    T = np.asarray(T_w2c)
    if T.shape == (4, 4):
        return T[:3, :3], T[:3, 3]
    if T.shape == (3, 4):
        return T[:, :3], T[:, 3]
    raise ValueError(f"T_w2c must be (3,4) or (4,4), got {T.shape}")


def _view_from_cv(pts, R_p3d, T_p3d):
    """Row-vector convention Xv = Xw @ R_p3d + T_p3d, order ((x*R0j + y*R1j) + z*R2j) + Tj."""
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    out = np.empty_like(pts, dtype=F32)
    for j in range(3):
        out[:, j] = ((x * R_p3d[0, j] + y * R_p3d[1, j]) + z * R_p3d[2, j]) + T_p3d[j]
    return out


def pinhole_camera_params(R_cv, t_cv, K, H: int, W: int):
    """cameras_from_opencv_projection: (R_p3d (3,3), T_p3d (3,), focal (2,), principal point (2,)) in float32."""
    R = np.asarray(R_cv, dtype=F32)
    t = np.asarray(t_cv, dtype=F32)
    K = np.asarray(K, dtype=F32)
    scale = F32(min(W, H)) / F32(2.0)
    c0 = np.array([F32(W) / F32(2.0), F32(H) / F32(2.0)], dtype=F32)
    focal = np.array([K[0, 0], K[1, 1]], dtype=F32) / scale
    pp = -(np.array([K[0, 2], K[1, 2]], dtype=F32) - c0) / scale
    R_p3d = R.T.copy()
    R_p3d[:, :2] *= F32(-1)
    T_p3d = t.copy()
    T_p3d[:2] *= F32(-1)
    return R_p3d, T_p3d, focal.astype(F32), pp.astype(F32)


def project_pinhole(pts, R_cv, t_cv, K, H: int, W: int) -> np.ndarray:
    """(P,3) float32: NDC x, NDC y, view depth."""
    pts = np.asarray(pts, dtype=F32)
    R_p3d, T_p3d, focal, pp = pinhole_camera_params(R_cv, t_cv, K, H, W)
    v = _view_from_cv(pts, R_p3d, T_p3d)
    out = np.empty_like(v)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        out[:, 0] = (v[:, 0] * focal[0] + v[:, 2] * pp[0]) / v[:, 2]
        out[:, 1] = (v[:, 1] * focal[1] + v[:, 2] * pp[1]) / v[:, 2]
    out[:, 2] = v[:, 2]
    return out


ARIA_DEFAULT_RADIAL = (-0.02340373583137989, 0.09388021379709244, -0.06088035926222801,
                       0.0053304750472307205, 0.003342868760228157, -0.0006356257363222539)
ARIA_DEFAULT_TANGENTIAL = (0.0005087381578050554, -0.0004747129278257489)
ARIA_DEFAULT_THIN_PRISM = (-0.0011330085108056664, -0.00025734835071489215,
                           0.00009328465239377692, 0.00009424977179151028)


def fisheye_camera_params(R_cv, t_cv, f_px, cx_px, cy_px, H: int, W: int):
    """(R_p3d, T_p3d, f_ndc, cx_ndc, cy_ndc): the reference's axis swap for the Aria camera and its pixel->NDC
    conversion of the intrinsics (python float arithmetic, then float32)."""
    R = np.asarray(R_cv, dtype=F32)
    t = np.asarray(t_cv, dtype=F32)
    C = (np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]], dtype=F32)
         @ np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]], dtype=F32))
    R_p3d = (R.T @ C).astype(F32)
    T_p3d = (C.T @ t).astype(F32)
    f_ndc = 2.0 * f_px / W
    cx_ndc = 2.0 * (cx_px / W) - 1.0
    cy_ndc = 1.0 - 2.0 * (cy_px / H)
    return R_p3d, T_p3d, F32(f_ndc), F32(cx_ndc), F32(cy_ndc)


def project_fisheye(pts, R_cv, t_cv, f_px, cx_px, cy_px, radial, tangential, thin_prism, H: int, W: int):
    """FishEyeCameras(world_coordinates=True).transform_points, z replaced by view depth."""
    pts = np.asarray(pts, dtype=F32)
    R_p3d, T_p3d, f_ndc, cx_ndc, cy_ndc = fisheye_camera_params(R_cv, t_cv, f_px, cx_px, cy_px, H, W)
    k = np.asarray(radial, dtype=F32)
    p = np.asarray(tangential, dtype=F32)
    sp = np.asarray(thin_prism, dtype=F32)
    v = _view_from_cv(pts, R_p3d, T_p3d)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        a = v[:, 0] / v[:, 2]
        b = v[:, 1] / v[:, 2]
        r = np.sqrt(a * a + b * b)
        th = np.arctan(r).astype(F32)
        t1 = th * th
        t2 = t1 * t1
        t3 = t2 * t1
        t4 = t3 * t1
        t5 = t4 * t1
        t6 = t5 * t1
        th_radial = F32(1) + (((((k[0] * t1 + k[1] * t2) + k[2] * t3) + k[3] * t4) + k[4] * t5) + k[5] * t6)
        th_divr = np.where(np.abs(r) < F32(1e-10), F32(1), th / r)
        coeff = th_radial * th_divr
        xr = coeff * a
        yr = coeff * b
        n2 = xr * xr + yr * yr
        temp = F32(2.0) * (xr * p[0] + yr * p[1])
        ux = (xr + temp * xr) + n2 * p[0]
        uy = (yr + temp * yr) + n2 * p[1]
        n4 = n2 * n2
        ux = ux + (sp[0] * n2 + sp[1] * n4)
        uy = uy + (sp[2] * n2 + sp[3] * n4)
        out = np.empty_like(v)
        out[:, 0] = f_ndc * ux + cx_ndc
        out[:, 1] = f_ndc * uy + cy_ndc
    out[:, 2] = v[:, 2]
    return out


def radius_ndc(point_size, H: int, W: int) -> float:
    r = float(point_size) * (2.0 / max(W, H))
    return max(1e-5, min(0.25, r))


# --------------------------------------------------------------------------------------------------------------------
# R10 / R11 / R12
# --------------------------------------------------------------------------------------------------------------------
def pixel_ndc_centres(H: int, W: int):
    """NDC x of every column and NDC y of every row (float32), +X left, +Y up."""
    L = _lib()
    xf = np.array([L.oracle_pix_to_ndc(W - 1 - xi, W, H) for xi in range(W)], dtype=F32)
    yf = np.array([L.oracle_pix_to_ndc(H - 1 - yi, H, W) for yi in range(H)], dtype=F32)
    return xf, yf


def rasterize(ndc: np.ndarray, radius: float, H: int, W: int, K: int = 20, method: str = "naive", rows=None):
    """idx (H,W,K) int32, zbuf (H,W,K) f32, dists (H,W,K) f32; unfilled = -1."""
    L = _lib()
    pts = np.ascontiguousarray(ndc, dtype=F32).reshape(-1, 3)
    P = pts.shape[0]
    idx = np.full((H, W, K), -1, dtype=np.int32)
    zbuf = np.full((H, W, K), -1, dtype=F32)
    dists = np.full((H, W, K), -1, dtype=F32)
    r32 = F32(radius)
    if method == "naive":
        rad = np.full((max(P, 1),), r32, dtype=F32)
        y0, y1 = (0, H) if rows is None else rows
        L.oracle_rasterize_points_naive(_p(pts), _p(rad), P, H, W, K, y0, y1, _p(idx), _p(zbuf), _p(dists))
    elif method == "scatter":
        L.oracle_rasterize_points_scatter(_p(pts), float(r32), P, H, W, K, _p(idx), _p(zbuf), _p(dists))
    else:
        raise ValueError(method)
    return idx, zbuf, dists


def composite(idx, dists, feats, radius: float, background=(0.0, 0.0, 0.0)) -> np.ndarray:
    L = _lib()
    H, W, K = idx.shape
    feats = np.ascontiguousarray(feats, dtype=F32).reshape(-1, 3)
    if feats.shape[0] == 0:
        feats = np.zeros((1, 3), dtype=F32)
    out = np.empty((H, W, 3), dtype=F32)
    rr = F32(float(radius) * float(radius))  # double product, then float32
    bg = np.asarray(background, dtype=F32)
    L.oracle_alpha_composite(_p(np.ascontiguousarray(idx)), _p(np.ascontiguousarray(dists)), _p(feats), 3, H, W, K,
                             float(rr), _p(bg), _p(out))
    return out


def postprocess(img_f32: np.ndarray, rotate_cw: bool, bg_img: np.ndarray = None) -> np.ndarray:
    """scripts/render_vipe_pointcloud.py:879-888 (RGB branch): clamp, dark pixels take the background image's pixel
    (black when bg_img is None, the only background the reference's main ever passes), uint8 by truncation, rotation."""
    rgb = np.clip(img_f32, F32(0), F32(1))
    dark = np.all(rgb < F32(1e-2), axis=2)
    rgb = rgb.copy()
    rgb[dark] = F32(0) if bg_img is None else np.asarray(bg_img, dtype=F32)[dark]
    u8 = (np.clip(rgb, F32(0), F32(1)) * F32(255)).astype(np.uint8)
    if rotate_cw:
        u8 = np.ascontiguousarray(np.rot90(u8, k=-1))  # out[i][j] = in[H-1-j][i]
    return u8


def render_points_pinhole(points_world, colors_world, K, T_w2c, W, H, point_size, method="scatter",
                          return_fragments=False):
    R, t = split_w2c(np.asarray(T_w2c))
    pts, cols = filter_points(points_world, colors_world)
    if pts.shape[0] == 0:
        img = np.zeros((H, W, 3), dtype=np.uint8)
        return (img, None) if return_fragments else img
    ndc = project_pinhole(pts, R, t, K, H, W)
    r = radius_ndc(point_size, H, W)
    idx, zbuf, dists = rasterize(ndc, r, H, W, 20, method)
    img = postprocess(composite(idx, dists, cols, r), rotate_cw=True)
    if return_fragments:
        return img, dict(ndc=ndc, idx=idx, zbuf=zbuf, dists=dists, cols=cols, radius=r)
    return img


def scale_intrinsics_single(f, cx, cy, original_size, new_size):
    oh, ow = original_size
    nh, nw = new_size
    sw, sh = nw / ow, nh / oh
    return f * sw, cx * sw, cy * sh


def render_points_fisheye(points_world, colors_world, T_w2c, f_px, cx_px, cy_px, radial, tangential, thin_prism,
                          W, H, point_size, method="scatter", return_fragments=False):
    R, t = split_w2c(np.asarray(T_w2c))
    pts, cols = filter_points(points_world, colors_world)
    if pts.shape[0] == 0:
        img = np.zeros((H, W, 3), dtype=np.uint8)
        return (img, None) if return_fragments else img
    ndc = project_fisheye(pts, R, t, f_px, cx_px, cy_px, radial, tangential, thin_prism, H, W)
    r = radius_ndc(point_size, H, W)
    idx, zbuf, dists = rasterize(ndc, r, H, W, 20, method)
    img = postprocess(composite(idx, dists, cols, r), rotate_cw=False)
    if return_fragments:
        return img, dict(ndc=ndc, idx=idx, zbuf=zbuf, dists=dists, cols=cols, radius=r)
    return img


# --------------------------------------------------------------------------------------------------------------------
# output side: white-pixel metric
# --------------------------------------------------------------------------------------------------------------------
def pil_luma(frame_u8) -> np.ndarray:
    """PIL's Image.convert("L") on an RGB image (ITU-R 601-2 luma in 16.16 fixed point, rounded):
    (19595 R + 38470 G + 7471 B + 0x8000) >> 16.  Pinned by tests/golden/g4_pil_luma.npz (made with PIL itself)."""
    a = np.asarray(frame_u8).astype(np.uint32)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def white_pixel_mask(frame_u8, threshold: float = 30.0, grey: str = "pil"):
    """data_preprocess/scripts/postprocess_render_dir.py:17-31.  grey="pil" (:17-19, the branch taken whenever PIL
    imports): grey = float32(Image.convert("L")); grey="float" (:20-25, the imageio fallback):
    grey = float32(0.299 r + 0.587 g + 0.114 b) with the products and sums in float64 (uint8 arrays times Python
    floats).  mask = grey >= threshold; returns (0/1 uint8 mask, 0/255 uint8 picture).  :60-63: the metric is
    mask.sum() per frame."""
    arr = np.asarray(frame_u8)
    r, g, b = arr[..., 0], arr[..., 1], arr[..., 2]
    if grey == "pil":
        gray = pil_luma(arr).astype(F32)
    else:
        gray = (0.299 * r + 0.587 * g + 0.114 * b).astype(F32)
    logical = (gray >= F32(threshold)).astype(np.uint8)
    return logical, (logical * 255).astype(np.uint8)
