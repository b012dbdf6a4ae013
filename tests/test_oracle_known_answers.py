"""CPU: hand-derived known answers for the part of the oracle that restates pytorch3d (parity unpinned by the
reference: it ships no vectors for this boundary), plus self-consistency of the two rasterizer formulations."""
import numpy as np
import pytest

from oracle import oracle as O

F32 = np.float32


def test_pixel_centres_square_and_non_square():
    xf, yf = O.pixel_ndc_centres(4, 4)
    assert np.allclose(xf, [0.75, 0.25, -0.25, -0.75]) and np.allclose(yf, [0.75, 0.25, -0.25, -0.75])
    # non-square: the longer axis spans +-max/min, the shorter +-1
    xf, yf = O.pixel_ndc_centres(2, 4)
    assert np.allclose(yf, [0.5, -0.5]) and np.allclose(xf, [1.5, 0.5, -0.5, -1.5])
    xf, yf = O.pixel_ndc_centres(4, 2)
    assert np.allclose(xf, [0.5, -0.5]) and np.allclose(yf, [1.5, 0.5, -0.5, -1.5])


def test_single_point_lands_on_the_expected_pixel():
    H = W = 8
    xf, yf = O.pixel_ndc_centres(H, W)
    ndc = np.array([[xf[2], yf[5], 3.0]], dtype=F32)
    idx, zbuf, dists = O.rasterize(ndc, 0.1, H, W, 20)
    assert idx[5, 2, 0] == 0 and zbuf[5, 2, 0] == 3.0 and dists[5, 2, 0] == 0.0
    assert (idx >= 0).sum() == 1
    img = O.composite(idx, dists, np.array([[0.2, 0.4, 0.6]], F32), 0.1)
    assert np.allclose(img[5, 2], [0.2, 0.4, 0.6]) and img.sum() == pytest.approx(1.2)


def test_two_overlapping_discs_alpha_blend():
    H = W = 8
    xf, yf = O.pixel_ndc_centres(H, W)
    r = 0.3
    # both cover pixel (3,3); nearer one is offset by d, farther one centred
    d = F32(0.1)
    ndc = np.array([[xf[3], yf[3], 5.0], [xf[3] + d, yf[3], 2.0]], dtype=F32)
    cols = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], F32)
    idx, zbuf, dists = O.rasterize(ndc, r, H, W, 20)
    assert list(idx[3, 3, :2]) == [1, 0]  # front to back
    rr = F32(r * r)
    dd = F32((ndc[1, 0] - xf[3]) ** 2)
    a1 = F32(1) - dd / rr
    img = O.composite(idx, dists, cols, r)
    assert img[3, 3, 1] == pytest.approx(float(a1), rel=1e-6)           # green = a1
    assert img[3, 3, 0] == pytest.approx(float(1 - a1), rel=1e-5)       # red   = (1-a1) * 1


def test_ties_boundary_and_culls():
    H = W = 8
    xf, yf = O.pixel_ndc_centres(H, W)
    r = F32(0.2)
    ndc = np.array([[xf[1], yf[1], 1.0], [xf[1], yf[1], 1.0],           # exact z tie -> lower index first
                    [xf[4] + r, yf[4], 1.0],                            # distance == radius: strict <, no hit on (4,4)
                    [xf[6], yf[6], -1.0],                               # behind the camera
                    [xf[6], yf[6], 0.0]], dtype=F32)                    # z == 0 stays
    idx, _, _ = O.rasterize(ndc, float(r), H, W, 20)
    assert list(idx[1, 1, :3]) == [0, 1, -1]
    assert 2 not in idx[4, 4]
    assert list(idx[6, 6, :2]) == [4, -1]


def test_more_than_k_hits_keeps_the_k_nearest_in_order():
    H = W = 4
    xf, yf = O.pixel_ndc_centres(H, W)
    z = np.linspace(5, 1, 30).astype(F32)
    ndc = np.stack([np.full(30, xf[2]), np.full(30, yf[1]), z], axis=1).astype(F32)
    idx, zbuf, _ = O.rasterize(ndc, 0.1, H, W, 20)
    assert list(idx[1, 2]) == list(range(29, 9, -1))
    assert np.all(np.diff(zbuf[1, 2]) > 0)


@pytest.mark.parametrize("H,W", [(24, 24), (12, 20), (20, 12)])
def test_scatter_equals_naive(H, W):
    rng = np.random.default_rng(H * 100 + W)
    n = 3000
    ndc = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(-1, 6, n)], axis=1).astype(F32)
    ndc[:20, 2] = 1.0  # ties
    ndc[20] = [np.nan, 0, 1]
    for r in (0.02, 0.11, 0.25):
        a = O.rasterize(ndc, r, H, W, 20, "naive")
        b = O.rasterize(ndc, r, H, W, 20, "scatter")
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_postprocess_rules():
    img = np.zeros((2, 3, 3), F32)
    img[0, 0] = [0.009, 0.009, 0.009]   # dark -> background
    img[0, 1] = [0.009, 0.02, 0.0]      # one channel above the threshold -> kept
    img[0, 2] = [1.7, -0.2, 0.999]      # clamp, truncation (0.999*255 = 254.7 -> 254)
    out = O.postprocess(img, rotate_cw=False)
    assert list(out[0, 0]) == [0, 0, 0] and list(out[0, 1]) == [2, 5, 0] and list(out[0, 2]) == [255, 0, 254]
    rot = O.postprocess(img, rotate_cw=True)
    assert rot.shape == (3, 2, 3)
    for i in range(3):
        for j in range(2):
            assert np.array_equal(rot[i, j], out[2 - 1 - j, i])  # out[i][j] = in[H-1-j][i]


def test_pinhole_projection_matches_pixel_formula():
    """x_ndc = 1 - 2u/S for a square image, u = fx*Xc/Zc + cx (pixel centres at xi + 0.5)."""
    S = 64
    K = np.array([[50.0, 0, 30.0], [0, 55.0, 33.0], [0, 0, 1]], F32)
    R = np.eye(3, dtype=F32)
    t = np.array([0.1, -0.2, 0.3], F32)
    pts = np.array([[0.3, 0.2, 2.0], [-0.5, 0.1, 4.0]], F32)
    ndc = O.project_pinhole(pts, R, t, K, S, S)
    Xc = pts + t
    u = K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2]
    v = K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]
    assert np.allclose(ndc[:, 0], 1 - 2 * u / S, atol=1e-5) and np.allclose(ndc[:, 1], 1 - 2 * v / S, atol=1e-5)
    assert np.allclose(ndc[:, 2], Xc[:, 2])


def test_fisheye_axis_swap_and_centre():
    """A point on the optical axis lands on the principal point; view axes are (Yc, -Xc, Zc)."""
    H = W = 448
    R, t = np.eye(3, dtype=F32), np.zeros(3, F32)
    ndc = O.project_fisheye(np.array([[0, 0, 2.0]], F32), R, t, 190.0, 224.0, 224.0, O.ARIA_DEFAULT_RADIAL,
                            O.ARIA_DEFAULT_TANGENTIAL, O.ARIA_DEFAULT_THIN_PRISM, H, W)
    assert np.allclose(ndc[0, :2], [0.0, 0.0], atol=1e-6) and ndc[0, 2] == 2.0
    ndc = O.project_fisheye(np.array([[0.2, 0.0, 2.0]], F32), R, t, 190.0, 224.0, 224.0, (0,) * 6, (0, 0), (0,) * 4,
                            H, W)
    # +Xc maps to -y_view: NDC y decreases (image row increases), NDC x stays at the centre
    assert abs(ndc[0, 0]) < 1e-6 and ndc[0, 1] < 0
    th = np.arctan(0.1)
    assert ndc[0, 1] == pytest.approx(-(2 * 190.0 / W) * th, rel=1e-5)


def test_defined_order_transforms_match_numpy_matmul():
    """The oracle's fixed-order R@p+t (float32) and [p,1]@T.T (float64) against numpy's BLAS products, which is what
    the reference calls (:405, :442-443): equal to within 2 ulp / 1e-15."""
    rng = np.random.default_rng(3)
    a = 0.3
    c2w = np.eye(4, dtype=F32)
    c2w[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], F32)
    c2w[:3, 3] = [0.5, -1.0, 2.0]
    p = (rng.standard_normal((1000, 3)) * 3).astype(F32)
    ref = (c2w[:3, :3] @ p.T + c2w[:3, 3:4]).T
    got = O.apply_c2w_f32(p, c2w)
    assert np.abs(got - ref).max() <= 4 * np.spacing(np.abs(ref).max())
    T = np.linalg.inv(c2w).astype(np.float64)
    ref64 = (T @ np.hstack([p.astype(np.float64), np.ones((1000, 1))]).T).T[:, :3]
    assert np.allclose(O.apply_T_f64(p, T), ref64, rtol=1e-14, atol=1e-14)
    # identity pose (the fixed-camera case): exact
    assert np.array_equal(O.apply_c2w_f32(p, np.eye(4, dtype=F32)), p)


def test_radius_rule():
    assert O.radius_ndc(5.0, 448, 448) == pytest.approx(5 * 2 / 448)
    assert O.radius_ndc(1000, 448, 448) == 0.25 and O.radius_ndc(0, 448, 448) == 1e-5
    r = O.radius_ndc(5.0, 448, 448)
    assert F32(r) * F32(r) == F32(4.9824617e-4) and F32(r * r) == F32(4.9824617e-4)


def test_white_pixel_mask_known_answers():
    """Grey weights sum to 1, so a uniform colour v has grey v (up to float64 rounding, then float32): 30 passes a
    threshold of 30, 29 does not; pure red needs 101 (0.299 * 101 = 30.199)."""
    f = np.zeros((2, 3, 3), dtype=np.uint8)
    f[0, 0] = 30
    f[0, 1] = 29
    f[0, 2] = [101, 0, 0]
    f[1, 0] = [100, 0, 0]
    f[1, 1] = [0, 52, 0]   # 0.587 * 52 = 30.524
    f[1, 2] = [0, 0, 255]  # 0.114 * 255 = 29.07
    logical, vis = O.white_pixel_mask(f, 30.0, grey="float")
    assert logical.tolist() == [[1, 0, 1], [0, 1, 0]]
    assert vis.max() == 255 and vis.dtype == np.uint8
    # PIL's integer luma rounds to nearest: [100, 0, 0] -> 29.9 -> 30 passes, [0, 0, 255] -> 29.07 -> 29 does not
    logical, _ = O.white_pixel_mask(f, 30.0)
    assert logical.tolist() == [[1, 0, 1], [1, 1, 0]]


def test_pil_luma_matches_pil():
    """The grey conversion the reference's post-processing script makes (PIL Image.convert("L")), against outputs of
    PIL itself committed as a fixture, and against PIL live where it is installed."""
    from pathlib import Path
    g = np.load(Path(__file__).parent / "golden" / "g4_pil_luma.npz")
    assert np.array_equal(O.pil_luma(g["rgb"]), g["L"])
    try:
        from PIL import Image
    except Exception:
        return
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    assert np.array_equal(O.pil_luma(a), np.asarray(Image.fromarray(a, "RGB").convert("L")))


def test_config_a_plumbing_on_the_cpu():
    """BASELINE configs[0] is the reference's own CPU-runnable case (480x832 exo, subsample 2, pinhole ego 448x448):
    the oracle runs it end to end -- unprojection of two frames, filters, projection, the naive per-pixel rasterizer on a
    band of rows against the point-driven one, compositor, post-processing."""
    from egoprior import synth
    F32 = np.float32
    H, W = 480, 832
    d, rgb, m = [t.numpy() for t in synth.make_frames(H, W, [0, 24], 49)]
    intr = synth.exo_intrinsics(H, W)
    exo = synth.exo_w2c()
    T = np.linalg.inv(exo)
    parts = [O.unproject_frame(d[f], rgb[f], m[f], intr, np.eye(4, dtype=F32), T, 2, "static") for f in range(2)]
    dp, dc = O.unproject_frame(d[1], rgb[1], m[1], intr, np.eye(4, dtype=F32), T, 2, "dynamic")
    pts = np.concatenate([p[0] for p in parts] + [dp])
    cols = np.concatenate([p[1] for p in parts] + [dc])
    assert 150_000 < pts.shape[0] <= 2 * 240 * 416 and dp.shape[0] > 0
    K_ego = np.array([[150.0, 0, 224.0], [0, 150.0, 224.0], [0, 0, 1]], dtype=F32)
    R, t = np.asarray(synth.ego_orbit_w2c(49, exo)[24], dtype=F32)[:3, :3], np.asarray(synth.ego_orbit_w2c(49, exo)[24], dtype=F32)[:3, 3]
    p32, c01 = O.filter_points(pts, cols)
    ndc = O.project_pinhole(p32, R, t, K_ego, 448, 448)
    r = O.radius_ndc(1.0, 448, 448)
    idx_s, z_s, d_s = O.rasterize(ndc, r, 448, 448, 20, "scatter")
    idx_n, z_n, d_n = O.rasterize(ndc, r, 448, 448, 20, "naive", rows=(220, 228))
    assert np.array_equal(idx_n[220:228], idx_s[220:228]) and np.array_equal(z_n[220:228], z_s[220:228])
    assert np.array_equal(d_n[220:228], d_s[220:228])
    img = O.postprocess(O.composite(idx_s, d_s, c01, r), rotate_cw=True)
    assert img.shape == (448, 448, 3) and img.dtype == np.uint8 and img.any()
    whole = O.render_points_pinhole(pts, cols, K_ego, np.asarray(synth.ego_orbit_w2c(49, exo)[24], dtype=F32), 448, 448, 1.0)
    assert np.array_equal(whole, img)


def test_postprocess_with_a_background_image():
    """scripts/render_vipe_pointcloud.py:879-885: a pixel is background when all three clamped channels are below 1e-2,
    and then takes the background image's pixel; uint8 by truncation."""
    img = np.array([[[0.0, 0.0, 0.0], [0.009, 0.0099, 0.0], [0.5, 0.0, 0.0]],
                    [[0.01, 0.0, 0.0], [-3.0, 0.005, 0.0], [2.0, 0.999, 0.0039]]], dtype=np.float32)
    bg = np.full((2, 3, 3), 0.5, dtype=np.float32)
    bg[0, 1] = (1.5, -0.2, 0.9999)
    out = O.postprocess(img, False, bg_img=bg)
    assert out.tolist() == [[[127, 127, 127], [255, 0, 254], [127, 0, 0]],
                            [[2, 0, 0], [127, 127, 127], [255, 254, 0]]]
    assert O.postprocess(img, False).tolist() == [[[0, 0, 0], [0, 0, 0], [127, 0, 0]],
                                                  [[2, 0, 0], [0, 0, 0], [255, 254, 0]]]
    rot = O.postprocess(img, True, bg_img=bg)
    assert rot.shape == (3, 2, 3) and rot[0, 1].tolist() == [127, 127, 127] and rot[2, 0].tolist() == [255, 254, 0]
