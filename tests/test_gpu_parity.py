"""GPU parity: every HIP stage, called through the C ABI, against the CPU oracle on identical inputs.

Tolerances (the bar): index maps, validity masks, point counts/order, colours and uint8 frames are compared bit-exact;
floating-point outputs are compared bit-exact too wherever the oracle fixes the operation order (everything except the
fisheye projection, whose atan differs between libm and the GPU math library: 1e-5 relative there).
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

F32 = np.float32


def pack_ndc(ndc, cols_u8, device):
    n = ndc.shape[0]
    rec = np.zeros((max(n, 1), 4), dtype=F32)
    rec[:n, :3] = ndc
    c = np.asarray(cols_u8, dtype=np.uint32).reshape(-1, 3)
    rgba = (c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16)).astype(np.uint32)
    rec[:n, 3] = rgba.view(F32)
    return torch.from_numpy(rec).to(device)


def synthetic_clip(H=120, W=208, F=3):
    from egoprior import synth
    d, rgb, m = synth.make_frames(H, W, range(F), F)
    E = synth.exo_w2c()
    T = np.linalg.inv(E)
    return dict(depth=d.numpy(), rgb=rgb.numpy(), mask=m.numpy(), intr=synth.exo_intrinsics(H, W),
                c2w=np.eye(4, dtype=F32), T=T, E=E, egos=synth.ego_orbit_w2c(F, E))


# ----------------------------------------------------------------------------------------------------------------------
def test_reliable_mask_matches_oracle_and_golden(engine):
    from pathlib import Path
    g = np.load(Path(__file__).parent / "golden" / "g1_rays_mask.npz")
    for key_d, key_m in (("depth", "mask"), ("depth2", "mask2")):
        got = engine.reliable_depth_mask(g[key_d]).cpu().numpy()
        assert np.array_equal(got, g[key_m])
    clip = synthetic_clip(135, 241, 1)  # odd sizes: partial blocks on both axes
    got = engine.reliable_depth_mask(clip["depth"][0]).cpu().numpy()
    assert np.array_equal(got, O.reliable_depth_mask_range(clip["depth"][0]))
    rng = np.random.default_rng(0)
    d = (rng.random((77, 130), dtype=F32) * 0.1 + 2.0).astype(F32)
    d[rng.random(d.shape) < 0.02] = np.nan
    d[rng.random(d.shape) < 0.02] = 0.0
    d[rng.random(d.shape) < 0.01] = np.inf
    d[rng.random(d.shape) < 0.01] = -1.0
    got = engine.reliable_depth_mask(d).cpu().numpy()
    assert np.array_equal(got, O.reliable_depth_mask_range(d))


@pytest.mark.parametrize("s", [1, 2, 3, 5])
@pytest.mark.parametrize("select", ["static", "dynamic"])
def test_unproject_frame(engine, s, select):
    from egoprior import _lib
    clip = synthetic_clip(120, 208, 2)
    sel = _lib.SELECT_STATIC if select == "static" else _lib.SELECT_DYNAMIC
    # a non-trivial exo pose so that the float32 c2w product is exercised
    c2w = np.eye(4, dtype=F32)
    a = 0.1
    c2w[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=F32)
    c2w[:3, 3] = [0.1, -0.2, 0.3]
    for c in (clip["c2w"], c2w):
        ref_p, ref_c = O.unproject_frame(clip["depth"][1], clip["rgb"][1], clip["mask"][1], clip["intr"], c, clip["T"],
                                         s, select)
        cloud = engine.unproject_frame(clip["depth"][1], clip["rgb"][1], clip["mask"][1], clip["intr"], c, clip["T"],
                                       subsample=s, select=sel, want_xyz64=True)
        n = cloud.n()
        assert n == ref_p.shape[0]
        p32, cols = cloud.to_numpy()
        assert np.array_equal(cols, ref_c)
        assert np.array_equal(cloud.xyz64[:n].cpu().numpy(), ref_p)  # float64 exo->world, bit-exact
        assert np.array_equal(p32, ref_p.astype(F32))
        assert cloud.flags.item() == int(ref_c.size > 0 and ref_c.max() > 1)


@pytest.mark.parametrize("s", [1, 2, 3, 5])
def test_float16_depth_gives_the_same_bits(engine, s):
    """Depth handed over as float16 (the precision of the EXR artifacts) is widened on the device; masks, points and the
    mean background must equal what the float32 path gives for the same values, and the oracle's result."""
    from egoprior import _lib
    clip = synthetic_clip(120, 208, 6)  # make_frames rounds its depth to float16 values
    d32 = clip["depth"]
    d16 = d32.astype(np.float16)
    assert np.array_equal(d16.astype(F32), d32)
    m32 = engine.reliable_depth_mask(d32[0])
    m16 = engine.reliable_depth_mask(d16[0])
    assert torch.equal(m16, m32)
    assert np.array_equal(m16.cpu().numpy(), O.reliable_depth_mask_range(d32[0]))
    for select, name in ((_lib.SELECT_STATIC, "static"), (_lib.SELECT_DYNAMIC, "dynamic")):
        a = engine.unproject_frame(d32[1], clip["rgb"][1], clip["mask"][1], clip["intr"], clip["c2w"], clip["T"],
                                   subsample=s, select=select, want_xyz64=True)
        b = engine.unproject_frame(d16[1], clip["rgb"][1], clip["mask"][1], clip["intr"], clip["c2w"], clip["T"],
                                   subsample=s, select=select, want_xyz64=True)
        n = a.n()
        assert n == b.n() and n > 0
        assert torch.equal(a.points[:n], b.points[:n]) and torch.equal(a.xyz64[:n], b.xyz64[:n])
        ref_p, _ = O.unproject_frame(d32[1], clip["rgb"][1], clip["mask"][1], clip["intr"], clip["c2w"], clip["T"], s,
                                     name)
        assert np.array_equal(b.xyz64[:n].cpu().numpy(), ref_p)
    if s == 1:
        st32, st16 = engine.meanbg_new_state(120, 208), engine.meanbg_new_state(120, 208)
        engine.meanbg_accumulate(st32, d32, clip["rgb"], clip["mask"], 0)
        engine.meanbg_accumulate(st16, torch.from_numpy(d16).to(engine.device), clip["rgb"], clip["mask"], 0)
        assert torch.equal(st16, st32)


def test_unproject_without_mask(engine):
    from egoprior import _lib
    clip = synthetic_clip(64, 96, 1)
    ref_p, ref_c = O.unproject_frame(clip["depth"][0], clip["rgb"][0], None, clip["intr"], clip["c2w"], clip["T"], 2,
                                     "static")
    cloud = engine.unproject_frame(clip["depth"][0], clip["rgb"][0], None, clip["intr"], clip["c2w"], clip["T"],
                                   subsample=2, select=_lib.SELECT_STATIC)
    p32, cols = cloud.to_numpy()
    assert np.array_equal(p32, ref_p.astype(F32)) and np.array_equal(cols, ref_c)
    cloud = engine.unproject_frame(clip["depth"][0], clip["rgb"][0], None, clip["intr"], clip["c2w"], clip["T"],
                                   subsample=2, select=_lib.SELECT_DYNAMIC)
    assert cloud.n() == 0  # no mask file => no dynamic points (reference :728-729)


def test_filter_cloud(engine):
    rng = np.random.default_rng(1)
    pts = (rng.standard_normal((5000, 3)) * 3).astype(F32)
    pts[10] = [np.nan, 0, 0]
    pts[11] = [0, np.inf, 0]
    pts[12] = [2e6, 0, 0]
    pts[13] = [6e5, 6e5, 6e5]  # norm > 1e6
    pts[14] = [5e5, 5e5, 5e5]  # norm < 1e6
    pts[4999] = [0, 0, -np.inf]
    cols = rng.integers(0, 256, (5000, 3), dtype=np.uint8)
    ref_p, ref_c = O.filter_points(pts, cols)
    cloud = engine.filter_cloud(pts, cols)
    p32, c8 = cloud.to_numpy()
    assert np.array_equal(p32, ref_p)
    assert np.array_equal(c8.astype(F32) / F32(255), ref_c)
    assert cloud.flags.item() == 1
    # colours that never exceed 1 are not rescaled (reference :833-834)
    cols01 = rng.integers(0, 2, (100, 3), dtype=np.uint8)
    cloud = engine.filter_cloud(pts[100:200], cols01)
    assert cloud.flags.item() == 0
    empty = engine.filter_cloud(np.zeros((0, 3), F32), np.zeros((0, 3), np.uint8))
    assert empty.n() == 0


def _cams(clip, He, We):
    from egoprior import engine as eng, synth
    K = synth.ego_pinhole_K(He, We)
    R, t = O.split_w2c(clip["egos"][0])
    fx = synth.load_camera_fixture()
    prm = fx["online_calibration_line0"]["CameraCalibrations"][0]["Projection"]["Params"]
    f, cx, cy = O.scale_intrinsics_single(prm[0], prm[1], prm[2], (2880, 2880), (He, We))
    return K, R, t, (f, cx, cy, prm[3:9], prm[9:11], prm[11:15]), eng


@pytest.mark.parametrize("size", [(96, 160), (128, 128), (160, 96)])
def test_project_points(engine, size):
    He, We = size
    clip = synthetic_clip(120, 208, 1)
    K, R, t, fe, eng = _cams(clip, He, We)
    pts, cols = O.unproject_frame(clip["depth"][0], clip["rgb"][0], clip["mask"][0], clip["intr"], clip["c2w"],
                                  clip["T"], 1, "static")
    # points behind / on the camera plane too
    C = -R.T @ t
    extra = np.stack([C + R.T @ np.array([0.1, 0.1, -1.0]), C + R.T @ np.array([0.3, -0.2, 0.0])])
    pts = np.vstack([pts, extra])
    cols = np.vstack([cols, np.zeros((2, 3), np.uint8)])
    cloud = engine.filter_cloud(pts.astype(F32), cols)
    p32, _ = cloud.to_numpy()
    ref = O.project_pinhole(p32, R, t, K, He, We)
    got = engine.project_points(eng.pinhole_camera(R, t, K, He, We), cloud)[:cloud.n(), :3].cpu().numpy()
    assert np.array_equal(got, ref, equal_nan=True)  # fixed operation order, no FMA: bit-exact
    f, cx, cy, rad, tan, thin = fe
    ref = O.project_fisheye(p32, R, t, f, cx, cy, rad, tan, thin, He, We)
    got = engine.project_points(eng.fisheye_camera(R, t, f, cx, cy, rad, tan, thin, He, We), cloud)
    got = got[:cloud.n(), :3].cpu().numpy()
    ok = np.isfinite(ref).all(axis=1)
    assert np.array_equal(np.isfinite(got).all(axis=1), ok)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-5, atol=1e-6)  # atanf: GPU math library vs libm
    assert np.array_equal(got[:, 2], ref[:, 2], equal_nan=True)


@pytest.fixture(params=[0, 1, 2, 3, 4, 5], ids=["sweep448", "bubble", "sweep64", "sweep1024", "sweep384", "sweep512"])
def splat_variant(request):
    """Every splat kernel variant must give the same bits: the default sweep kernel (448-record chunks: seven records per
    lane, not a power of two), the plain register-insertion kernel, the sweep kernel with 64-record chunks (forces its
    multi-chunk merge on these small inputs), with 1024-record chunks, with 384-record chunks (four waves per SIMD) and
    with 512-record chunks."""
    from egoprior import _lib
    _lib.set_option("splat_variant", request.param)
    yield request.param
    _lib.set_option("splat_variant", 0)


@pytest.fixture(params=[(1, 1), (0, 1), (1, 0)], ids=["count-lds", "count-global", "scan-loop"])
def bin_variant(request):
    """Counting aggregated in LDS (default) or straight on the global counters; the tile scan from registers after one
    round of loads (default) or in its looping form."""
    from egoprior import _lib
    _lib.set_option("bin_lds", request.param[0])
    _lib.set_option("scan_once", request.param[1])
    yield request.param
    _lib.set_option("bin_lds", 1)
    _lib.set_option("scan_once", 1)


def _check_fragments(res, ref_idx, ref_z, ref_d, K):
    assert np.array_equal(res["idx"].cpu().numpy(), ref_idx[..., :K])
    assert np.array_equal(res["zbuf"].cpu().numpy(), ref_z[..., :K])
    assert np.array_equal(res["dists"].cpu().numpy(), ref_d[..., :K])


@pytest.mark.parametrize("size,ps", [((64, 64), 1.0), ((96, 160), 3.0), ((160, 96), 5.0), ((50, 70), 8.0),
                                     ((40, 40), 30.0)])
def test_rasterize_composite_on_identical_ndc(engine, size, ps, splat_variant, bin_variant):
    from egoprior import engine as eng
    He, We = size
    clip = synthetic_clip(120, 208, 1)
    K, R, t, _, _ = _cams(clip, He, We)
    pts, cols = O.unproject_frame(clip["depth"][0], clip["rgb"][0], None, clip["intr"], clip["c2w"], clip["T"], 1,
                                  "static")
    p32, c01 = O.filter_points(pts, cols)
    ndc = O.project_pinhole(p32, R, t, K, He, We)
    r = O.radius_ndc(ps, He, We)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, He, We, 20, "scatter")
    ref_img = O.composite(ref_idx, ref_d, c01, r)
    for rot in (False, True):
        rp = eng.make_raster_params(ps, He, We, K=20, rotate_cw=rot, colors_gt1=1)
        res = engine.rasterize_composite(pack_ndc(ndc, cols, engine.device), None, rp, He, We, want_f32=True,
                                         want_fragments=True)
        assert res["status"].item() == 0
        _check_fragments(res, ref_idx, ref_z, ref_d, 20)
        assert np.array_equal(res["image_f32"].cpu().numpy(), ref_img)
        assert np.array_equal(res["image"].cpu().numpy(), O.postprocess(ref_img, rotate_cw=rot))
    # smaller K = the first K slots
    rp = eng.make_raster_params(ps, He, We, K=5, colors_gt1=1)
    res = engine.rasterize_composite(pack_ndc(ndc, cols, engine.device), None, rp, He, We, want_fragments=True)
    _check_fragments(res, ref_idx, ref_z, ref_d, 5)


def test_rasterize_known_answers_and_ties(engine, splat_variant):
    """Hand-built cases: z ties resolved by index, a point exactly on the radius boundary, z < 0, z = 0, NaN, more than
    K hits on one pixel, points outside the image."""
    from egoprior import engine as eng
    H = W = 16
    ps = 2.0
    r = O.radius_ndc(ps, H, W)
    xf, yf = O.pixel_ndc_centres(H, W)
    pts = []
    for k in range(30):  # 30 coincident discs on pixel (5, 7) with equal z: the 20 lowest indices must win, in order
        pts.append([xf[7], yf[5], 2.0])
    pts.append([xf[3], yf[3], 1.0])
    pts.append([xf[3], yf[3], 1.0])  # exact tie pair
    pts.append([xf[10] + F32(r), yf[10], 1.5])  # centre-to-pixel distance == r: strict < fails for pixel (10,10)
    pts.append([xf[12], yf[12], -0.5])  # behind
    pts.append([xf[12], yf[12], 0.0])  # z == 0 is kept
    pts.append([xf[13], yf[13], -0.0])
    pts.append([np.nan, yf[1], 1.0])
    pts.append([xf[1], np.inf, 1.0])
    pts.append([5.0, 5.0, 1.0])  # far outside
    pts.append([xf[0] + F32(0.5) * F32(r), yf[0], 3.0])  # partly outside the image
    ndc = np.asarray(pts, dtype=F32)
    rng = np.random.default_rng(2)
    cols = rng.integers(0, 256, (ndc.shape[0], 3), dtype=np.uint8)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, H, W, 20, "naive")
    assert list(ref_idx[5, 7]) == list(range(20))
    assert list(ref_idx[3, 3][:2]) == [30, 31]
    rp = eng.make_raster_params(ps, H, W, K=20, colors_gt1=1)
    res = engine.rasterize_composite(pack_ndc(ndc, cols, engine.device), None, rp, H, W, want_f32=True,
                                     want_fragments=True)
    _check_fragments(res, ref_idx, ref_z, ref_d, 20)
    ref_img = O.composite(ref_idx, ref_d, cols.astype(F32) / F32(255), r)
    assert np.array_equal(res["image_f32"].cpu().numpy(), ref_img)


def _ulps(v, n):
    """float32 v moved by n units in the last place."""
    v = F32(v)
    for _ in range(abs(n)):
        v = np.nextafter(v, F32(np.inf if n > 0 else -np.inf), dtype=F32)
    return v


@pytest.mark.parametrize("size,ps", [((40, 56), 1.0), ((40, 56), 2.5), ((56, 40), 5.0), ((24, 24), 9.0)])
def test_rasterize_discs_that_touch_pixel_centres_exactly(engine, size, ps, splat_variant, bin_variant):
    """The binning keeps, per axis, the pixel centres in [c - r, c + r] and drops discs that fall between centres; the
    splat kernel drops list entries that cannot reach its tile.  Both are conservative by a margin, and this is the case
    that would show a margin that is too small: discs whose rim lies on a pixel centre, one ulp inside and one ulp
    outside, on either side of tile borders and of the image border, plus discs centred exactly between pixel centres."""
    from egoprior import engine as eng
    He, We = size
    r = F32(O.radius_ndc(ps, He, We))
    xf, yf = O.pixel_ndc_centres(He, We)
    pts = []
    cols_of_interest = [0, 1, 7, 8, 9, 15, 16, We - 2, We - 1]
    rows_of_interest = [0, 1, 7, 8, 9, 15, 16, He - 2, He - 1]
    z = 1.0
    for xi in cols_of_interest:
        for yi in rows_of_interest:
            cx, cy = xf[xi], yf[yi]
            for sx, sy in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1)):
                # a disc whose rim passes through this pixel centre, along an axis or a diagonal
                k = F32(1.0) if sx == 0 or sy == 0 else F32(np.sqrt(0.5))
                bx, by = F32(cx + sx * r * k), F32(cy + sy * r * k)
                for ulps in (0, 1, -1, 4, -4):
                    px = bx if sx == 0 else _ulps(bx, ulps)
                    py = by if sy == 0 else _ulps(by, ulps)
                    z += 0.001
                    pts.append((px, py, z))
    # discs centred midway between pixel centres (small radii reach nothing, larger ones reach both neighbours)
    for xi in (3, 7, 8, We - 2):
        for yi in (3, 7, 8, He - 2):
            z += 0.001
            pts.append((F32((xf[xi] + xf[xi + 1]) / 2), F32((yf[yi] + yf[yi + 1]) / 2), z))
    ndc = np.array(pts, dtype=F32)
    rng = np.random.default_rng(7)
    cols = rng.integers(2, 256, size=(ndc.shape[0], 3), dtype=np.uint8)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, He, We, 20, "scatter")
    assert (ref_idx >= 0).any()
    rp = eng.make_raster_params(ps, He, We, K=20, colors_gt1=1)
    res = engine.rasterize_composite(pack_ndc(ndc, cols, engine.device), None, rp, He, We, want_fragments=True)
    assert res["status"].item() == 0
    _check_fragments(res, ref_idx, ref_z, ref_d, 20)


@pytest.mark.parametrize("size", [(8, 8), (8, 16), (24, 8), (16, 16), (9, 17)])
@pytest.mark.parametrize("ps", [1.0, 3.0, 6.0, 11.0])
def test_rasterize_on_grids_narrower_than_a_disc(engine, size, ps, bin_variant):
    """Images of one to three tiles per axis, where the most tiles a disc can take on an axis is cut off by the grid: the
    binning's straight-line walks (rectangles of at most 2 x 2 and 3 x 3 tiles) and its per-lane loops all see rectangles
    that the image border clips, and candidate tiles beyond the last row or column must never be touched."""
    from egoprior import engine as eng
    He, We = size
    rng = np.random.default_rng(He * 100 + We + int(ps * 10))
    n = 3000
    ndc = np.empty((n, 3), dtype=F32)
    ndc[:, 0] = rng.uniform(-1.6, 1.6, n) * (We / min(He, We))
    ndc[:, 1] = rng.uniform(-1.6, 1.6, n) * (He / min(He, We))
    ndc[:, 2] = rng.uniform(0.5, 4.0, n)
    cols = rng.integers(2, 256, size=(n, 3), dtype=np.uint8)
    r = F32(O.radius_ndc(ps, He, We))
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, He, We, 20, "scatter")
    assert (ref_idx >= 0).any()
    rp = eng.make_raster_params(ps, He, We, K=20, colors_gt1=1)
    res = engine.rasterize_composite(pack_ndc(ndc, cols, engine.device), None, rp, He, We, want_fragments=True)
    assert res["status"].item() == 0
    _check_fragments(res, ref_idx, ref_z, ref_d, 20)


@pytest.mark.parametrize("mode", ["narrow_band", "one_depth_plus_outliers", "few_depths"])
def test_rasterize_clustered_depths(engine, mode, splat_variant):
    """Depth distributions that defeat equal-width bucketing: a narrow band of depths next to a wide spread, one single
    depth with two far outliers, and a handful of distinct depths.  These drive the sort stage through its long-bucket
    paths (wave-wide ranking, bitonic sort of the chunk); the fragments must still be exact."""
    from egoprior import engine as eng
    H = W = 32
    ps = 3.0
    n = 12000
    rng = np.random.default_rng({"narrow_band": 11, "one_depth_plus_outliers": 12, "few_depths": 13}[mode])
    xy = rng.uniform(-1.05, 1.05, (n, 2)).astype(F32)
    if mode == "narrow_band":
        z = np.where(rng.random(n) < 0.6, F32(5) + rng.uniform(0, 1e-5, n).astype(F32), rng.uniform(1, 50, n).astype(F32))
        z[rng.integers(0, n, 100)] = F32(1000)
    elif mode == "one_depth_plus_outliers":
        z = np.full(n, 3.25, F32)
        z[17], z[4242] = F32(0.5), F32(900)
    else:
        z = (F32(2) + rng.integers(0, 8, n).astype(F32) * np.spacing(F32(2))).astype(F32)
    ndc = np.concatenate([xy, z.astype(F32)[:, None]], axis=1).astype(F32)
    cols = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    r = O.radius_ndc(ps, H, W)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, H, W, 20, "scatter")
    rp = eng.make_raster_params(ps, H, W, K=20, colors_gt1=1)
    res = engine.rasterize_composite(pack_ndc(ndc, cols, engine.device), None, rp, H, W, want_f32=True,
                                     want_fragments=True)
    assert res["status"].item() == 0
    _check_fragments(res, ref_idx, ref_z, ref_d, 20)
    ref_img = O.composite(ref_idx, ref_d, cols.astype(F32) / F32(255), r)
    assert np.array_equal(res["image_f32"].cpu().numpy(), ref_img)


@pytest.mark.parametrize("mode", ["duplicates", "spread", "front_last"])
def test_rasterize_heavy_tiles_are_sliced(engine, mode, splat_variant):
    """A dense cloud over a few tiles: lists far longer than 32768 entries are cut into slices that several waves
    reduce and one of them merges, and most keys never pass the per-record thresholds once the pixel lists are full.
    duplicates: 24 exact copies of every point (the static cloud accumulated over a clip: equal depths, order by index);
    spread: independent depths; front_last: depths that decrease along the list, so that every new batch beats the lists
    so far.  The fragments must be the oracle's whichever way the work is cut, and the same with slicing switched off."""
    from egoprior import _lib, engine as eng
    H, W = 40, 48
    ps = 4.0
    rng = np.random.default_rng({"duplicates": 21, "spread": 22, "front_last": 23}[mode])
    n = 360_000
    if mode == "duplicates":
        base = n // 24
        xy0 = np.stack([rng.uniform(-0.55, 0.35, base), rng.uniform(-0.3, 0.45, base)], axis=1).astype(F32)
        z0 = rng.uniform(1.0, 4.0, base).astype(F32)
        xy, z = np.tile(xy0, (24, 1)), np.tile(z0, 24)
    else:
        xy = np.stack([rng.uniform(-0.55, 0.35, n), rng.uniform(-0.3, 0.45, n)], axis=1).astype(F32)
        z = rng.uniform(1.0, 4.0, n).astype(F32)
        if mode == "front_last":
            z = np.sort(z)[::-1].copy()
    ndc = np.concatenate([xy, z[:, None]], axis=1).astype(F32)
    cols = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    r = O.radius_ndc(ps, H, W)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, H, W, 20, "scatter")
    ref_img = O.composite(ref_idx, ref_d, cols.astype(F32) / F32(255), r)
    rp = eng.make_raster_params(ps, H, W, K=20, colors_gt1=1)
    packed = pack_ndc(ndc, cols, engine.device)
    for slices in (1, 0):
        _lib.set_option("slices", slices)
        try:
            res = engine.rasterize_composite(packed, None, rp, H, W, want_f32=True, want_fragments=True)
        finally:
            _lib.set_option("slices", 1)
        assert res["status"].item() == 0
        _check_fragments(res, ref_idx, ref_z, ref_d, 20)
        assert np.array_equal(res["image_f32"].cpu().numpy(), ref_img)


def test_rasterize_tail_tiles_are_sliced(engine, check_build):
    """Option "tail_slots": when the tiles with real work outnumber the wave slots and the last round of them would be at
    most half full, the tiles of that round are cut into three slices each (the hand-over of the heavy tiles, for lists of
    a few hundred entries).  64 tiles of 600 - 1400 entries with the option at 28 "slots": two full rounds and a tail of 8;
    at 50: one round and a tail of 14.  Same fragments and frame as the oracle's and as with the option off; the build
    with hand-over checksums confirms that the slices really ran (two hand-overs per tail tile) and read what was written."""
    from egoprior import _lib, engine as eng
    H = W = 64
    ps = 2.5
    rng = np.random.default_rng(31)
    n = 30_000
    xy = rng.uniform(-1.0, 1.0, (n, 2)).astype(F32)
    xy[:, 0] *= F32(0.98)
    z = (2.0 + xy[:, 0] * 0.5 + rng.uniform(0.0, 0.05, n)).astype(F32)  # a slanted sheet: depth order sweeps the tiles
    ndc = np.concatenate([xy, z[:, None]], axis=1).astype(F32)
    cols = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    r = O.radius_ndc(ps, H, W)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, H, W, 20, "scatter")
    ref_img = O.composite(ref_idx, ref_d, cols.astype(F32) / F32(255), r)
    rp = eng.make_raster_params(ps, H, W, K=20, colors_gt1=1)
    packed = pack_ndc(ndc, cols, engine.device)
    default_slots = _lib.lib().egoprior_get_option(b"tail_slots")
    for slots in (28, 50, 0):
        _lib.set_option("tail_slots", slots)
        try:
            res = engine.rasterize_composite(packed, None, rp, H, W, want_f32=True, want_fragments=True)
        finally:
            _lib.set_option("tail_slots", default_slots)
        assert res["status"].item() == 0
        _check_fragments(res, ref_idx, ref_z, ref_d, 20)
        assert np.array_equal(res["image_f32"].cpu().numpy(), ref_img)
    E2, L2 = check_build
    for slots, lo, hi in ((28, 2 * 8, 2 * 16), (50, 2 * 14, 2 * 24), (0, 0, 0)):
        L2.egoprior_set_option(b"tail_slots", slots)
        E2.check_counts()
        res = E2.rasterize_composite(packed, None, rp, H, W, want_f32=True, want_fragments=True)
        torch.cuda.synchronize()
        _check_fragments(res, ref_idx, ref_z, ref_d, 20)
        seen, failed = E2.check_counts()
        # the tail is the last round's tiles plus the longer members of the length class it starts in
        assert lo <= seen <= hi and failed == 0, (slots, seen, failed)


def test_sliced_tiles_with_exact_and_short_workspaces(engine):
    """A dense blob whose tiles are sliced, rendered in a workspace sized from the frame's own entry count (exact bits)
    and in one that holds only 40 % of the entries: the lists are cut, the status word says so, and nothing is read or
    written out of bounds (every slice's share of a cut list is empty or shortened)."""
    import ctypes as C
    from egoprior import _lib, engine as eng
    L, dev = engine.L, engine.device
    H, W, ps, K = 40, 48, 4.0, 20
    rng = np.random.default_rng(31)
    n = 300_000
    xy = np.stack([rng.uniform(-0.5, 0.3, n), rng.uniform(-0.3, 0.4, n)], axis=1).astype(F32)
    ndc = np.concatenate([xy, rng.uniform(1.0, 4.0, (n, 1)).astype(F32)], axis=1)
    cols = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    r = O.radius_ndc(ps, H, W)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, H, W, K, "scatter")
    rec = pack_ndc(ndc, cols, dev)
    rp = eng.make_raster_params(ps, H, W, K=K, colors_gt1=1)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws0 = torch.empty(L.egoprior_raster_workspace_bytes_entries(H, W, n, 0), dtype=torch.uint8, device=dev)
    entries = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(L.egoprior_count_list_entries(None, C.byref(rp), H, W, C.c_void_p(rec.data_ptr()), n, None, None, 0, None,
                                             C.c_void_p(entries.data_ptr()), C.c_void_p(ws0.data_ptr()), ws0.numel(), st))
    need = int(entries.item())
    assert need > 20 * 32768  # several tiles are long enough to be sliced

    def render(n_entries):
        ws = torch.empty(L.egoprior_raster_workspace_bytes_entries(H, W, n, n_entries), dtype=torch.uint8, device=dev)
        idx = torch.empty((H, W, K), dtype=torch.int32, device=dev)
        zb = torch.empty((H, W, K), dtype=torch.float32, device=dev)
        ds = torch.empty((H, W, K), dtype=torch.float32, device=dev)
        img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.egoprior_rasterize_composite(C.byref(rp), H, W, C.c_void_p(rec.data_ptr()), n, None, None, None,
                                                  C.c_void_p(img.data_ptr()), None, C.c_void_p(idx.data_ptr()),
                                                  C.c_void_p(zb.data_ptr()), C.c_void_p(ds.data_ptr()),
                                                  C.c_void_p(status.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), st))
        torch.cuda.synchronize()
        return idx.cpu().numpy(), zb.cpu().numpy(), ds.cpu().numpy(), int(status.item())

    idx, zb, ds, status = render(need)
    assert status == 0 and np.array_equal(idx, ref_idx) and np.array_equal(zb, ref_z) and np.array_equal(ds, ref_d)
    idx, zb, ds, status = render(need * 2 // 5)
    assert status == 1
    assert ((idx >= -1) & (idx < n)).all()                      # what was drawn is still made of real points
    hit = idx >= 0
    assert np.array_equal(zb[hit], ndc[idx[hit], 2])            # ... with their own depths


# EGOPRIOR_FUZZ_SEEDS=n runs n random geometries instead of 16 (a soak run for kernel changes)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("EGOPRIOR_FUZZ_SEEDS", "16")))))
def test_rasterize_random_geometry(engine, seed):
    """Random image sizes (ragged last tiles), point sizes from sub-pixel to a quarter of the image, K of 1, 5 or 20,
    and clouds that mix a uniform spread with a tight blob (every fourth seed: a blob dense enough for its tiles to be
    sliced) and depth ties: fragments and composite against the oracle."""
    from egoprior import engine as eng
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(9, 90)), int(rng.integers(9, 90))
    ps = float(rng.choice([0.4, 1.0, 2.0, 3.5, 6.0, 11.0]))
    K = int(rng.choice([1, 5, 20]))
    n_uniform = int(rng.integers(100, 20000))
    n_blob = 150_000 if seed % 4 == 3 else int(rng.integers(0, 30000))
    half_x, half_y = max(W / H, 1.0), max(H / W, 1.0)
    xy_u = np.stack([rng.uniform(-1.1 * half_x, 1.1 * half_x, n_uniform),
                     rng.uniform(-1.1 * half_y, 1.1 * half_y, n_uniform)], axis=1)
    c = np.array([rng.uniform(-0.6, 0.6) * half_x, rng.uniform(-0.6, 0.6) * half_y])
    xy_b = c + rng.normal(0.0, 0.08, (n_blob, 2))
    xy = np.concatenate([xy_u, xy_b]).astype(F32)
    z = rng.uniform(0.5, 9.0, xy.shape[0]).astype(F32)
    z[rng.random(z.shape[0]) < 0.05] = F32(2.5)           # ties, resolved by index
    z[rng.random(z.shape[0]) < 0.01] = F32(-1.0)          # behind the camera
    order = rng.permutation(xy.shape[0])
    ndc = np.concatenate([xy, z[:, None]], axis=1).astype(F32)[order]
    cols = rng.integers(0, 256, (ndc.shape[0], 3), dtype=np.uint8)
    r = O.radius_ndc(ps, H, W)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, H, W, K, "scatter")
    rp = eng.make_raster_params(ps, H, W, K=K, colors_gt1=1)
    res = engine.rasterize_composite(pack_ndc(ndc, cols, engine.device), None, rp, H, W, want_f32=True,
                                     want_fragments=True)
    assert res["status"].item() == 0
    _check_fragments(res, ref_idx, ref_z, ref_d, K)
    ref_img = O.composite(ref_idx, ref_d, cols.astype(F32) / F32(255), r)
    assert np.array_equal(res["image_f32"].cpu().numpy(), ref_img)


def test_rasterize_empty_and_all_behind(engine):
    from egoprior import engine as eng
    H, W = 40, 56
    rp = eng.make_raster_params(3.0, H, W, colors_gt1=1)
    ndc = np.array([[0.0, 0.0, -1.0], [0.1, 0.1, -2.0]], dtype=F32)
    res = engine.rasterize_composite(pack_ndc(ndc, np.full((2, 3), 255, np.uint8), engine.device), None, rp, H, W,
                                     want_fragments=True)
    assert (res["idx"].cpu().numpy() == -1).all() and (res["image"].cpu().numpy() == 0).all()
    live0 = torch.zeros(1, dtype=torch.int32, device=engine.device)
    res = engine.rasterize_composite(pack_ndc(ndc, np.full((2, 3), 255, np.uint8), engine.device), live0, rp, H, W)
    assert (res["image"].cpu().numpy() == 0).all()


@pytest.mark.parametrize("ps", [3.0, 20.0])
@pytest.mark.parametrize("fisheye", [False, True])
def test_render_frame_end_to_end(engine, fisheye, ps, splat_variant):
    """static + dynamic clouds built on the GPU from a synthetic frame, rendered from an ego pose; point_size 20 is the
    many-pixels-per-point case (BASELINE configs[3]: discs of 28 pixels across, every tile list several chunks long)."""
    from egoprior import _lib, engine as eng
    He, We = (112, 112) if fisheye else (96, 160)
    clip = synthetic_clip(120, 208, 2)
    K, R, t, fe, _ = _cams(clip, He, We)
    args = (clip["depth"][1], clip["rgb"][1], clip["mask"][1], clip["intr"], clip["c2w"], clip["T"])
    bg = engine.unproject_frame(*args, subsample=1, select=_lib.SELECT_STATIC)
    dyn = engine.unproject_frame(*args, subsample=1, select=_lib.SELECT_DYNAMIC)
    rp_ = O.unproject_frame(*args, 1, "static")
    rd_ = O.unproject_frame(*args, 1, "dynamic")
    pts = np.vstack([rp_[0], rd_[0]])
    cols = np.vstack([rp_[1], rd_[1]])
    if not fisheye:
        ref_img, fr = O.render_points_pinhole(pts, cols, K, clip["egos"][0], We, He, ps, return_fragments=True)
        cam = eng.pinhole_camera(R, t, K, He, We)
        rp = eng.make_raster_params(ps, He, We, rotate_cw=True)
        res = engine.render_frame(cam, rp, bg, dyn, want_fragments=True)
        assert res["status"].item() == 0
        _check_fragments(res, fr["idx"], fr["zbuf"], fr["dists"], 20)
        assert np.array_equal(res["image"].cpu().numpy(), ref_img)
        again = engine.render_frame(cam, rp, bg, dyn, want_fragments=True)
        assert torch.equal(again["image"], res["image"]) and torch.equal(again["idx"], res["idx"])  # deterministic
    else:
        f, cx, cy, rad, tan, thin = fe
        cam = eng.fisheye_camera(R, t, f, cx, cy, rad, tan, thin, He, We)
        rp = eng.make_raster_params(ps, He, We, rotate_cw=False)
        res = engine.render_frame(cam, rp, bg, dyn, want_f32=True, want_fragments=True)
        # the GPU's own NDC coordinates -> oracle rasterizer + compositor: everything downstream is bit-exact
        ndc = engine.project_points(cam, bg, dyn)[:bg.n() + dyn.n(), :3].cpu().numpy()
        r = O.radius_ndc(ps, He, We)
        ref_idx, ref_z, ref_d = O.rasterize(ndc, r, He, We, 20, "scatter")
        _check_fragments(res, ref_idx, ref_z, ref_d, 20)
        ref_img = O.composite(ref_idx, ref_d, cols.astype(F32) / F32(255), r)
        assert np.array_equal(res["image_f32"].cpu().numpy(), ref_img)
        assert np.array_equal(res["image"].cpu().numpy(), O.postprocess(ref_img, rotate_cw=False))
        # and against the all-CPU frame: only pixels whose coverage flips on a 1-ulp atan difference may differ
        full = O.render_points_fisheye(pts, cols, clip["egos"][0], f, cx, cy, rad, tan, thin, We, He, ps)
        diff = (res["image"].cpu().numpy().astype(int) - full.astype(int))
        assert (np.abs(diff).max(axis=2) > 1).mean() < 2e-3


@pytest.mark.parametrize("slots", [1, 3])
def test_frame_pipeline_matches_one_frame_at_a_time(engine, slots):
    """Frames in flight on several HIP streams give the same bits as render_frame called frame by frame, and the first
    frame equals the oracle's."""
    from egoprior import _lib
    from egoprior import engine as eng
    F, H, W, He, We = 7, 120, 208, 96, 160
    clip = synthetic_clip(H, W, F)
    dev = engine.device
    depth = torch.from_numpy(clip["depth"]).to(dev)
    rgb = torch.from_numpy(clip["rgb"]).to(dev)
    mask = torch.from_numpy(clip["mask"]).to(dev)
    bg = engine.unproject_frame(depth[0], rgb[0], mask[0], clip["intr"], clip["c2w"], clip["T"], subsample=1,
                                select=_lib.SELECT_STATIC)
    rp = eng.make_raster_params(4.0, He, We, K=20, rotate_cw=True, colors_gt1=-1)
    up = eng.make_unproject_params(H, W, 1, _lib.SELECT_DYNAMIC, clip["intr"], clip["c2w"], clip["T"])
    K = np.array([[0.8 * We, 0, We / 2], [0, 0.8 * We, He / 2], [0, 0, 1]], dtype=F32)
    cams = []
    for f in range(F):
        R, t = eng.split_w2c(clip["egos"][f])
        cams.append(eng.pinhole_camera(R, t, K, He, We))
    ref = []
    for f in range(F):
        dyn = engine.unproject_frame(depth[f], rgb[f], mask[f], clip["intr"], clip["c2w"], clip["T"], subsample=1,
                                     select=_lib.SELECT_DYNAMIC)
        ref.append(engine.render_frame(cams[f], rp, bg, dyn)["image"].cpu().numpy())
    out = torch.zeros((F, We, He, 3), dtype=torch.uint8, device=dev)
    pipe = eng.FramePipeline(engine, bg, rp, He, We, exo_hw=(H, W), subsample=1, slots=slots)
    for rep in range(2):  # the second round reuses every slot's buffers
        out.zero_()
        for f in range(F):
            pipe.submit(cams[f], out[f], up=up, depth=depth[f], rgb=rgb[f], mask=mask[f])
        pipe.finish()
        got = out.cpu().numpy()
        for f in range(F):
            assert np.array_equal(got[f], ref[f]), f"frame {f} differs (round {rep})"
    assert len({r.tobytes() for r in ref}) > 1  # the frames really differ from one another
    # frame 0 against the oracle
    sp, sc = O.unproject_frame(clip["depth"][0], clip["rgb"][0], clip["mask"][0], clip["intr"], clip["c2w"], clip["T"], 1,
                               "static")
    dp, dc = O.unproject_frame(clip["depth"][0], clip["rgb"][0], clip["mask"][0], clip["intr"], clip["c2w"], clip["T"], 1,
                               "dynamic")
    img = O.render_points_pinhole(np.vstack([sp, dp]), np.vstack([sc, dc]), K, clip["egos"][0], We, He, 4.0)
    assert np.array_equal(ref[0], img)


@pytest.mark.parametrize("F", [1, 5, 17, 40])
def test_mean_background(engine, F):
    from egoprior import _lib
    clip = synthetic_clip(72, 128, F)  # 72*128 is a multiple of 32 (see oracle.cascade_nansum)
    ref_d, ref_c = O.mean_background(clip["depth"], clip["rgb"], clip["mask"])
    st = engine.meanbg_new_state(72, 128)
    engine.meanbg_accumulate(st, clip["depth"], clip["rgb"], clip["mask"], 0)
    md, mc = engine.meanbg_finalize(st)
    assert np.array_equal(md.cpu().numpy(), ref_d)
    ref_q = np.clip(ref_c * F32(255), 0, 255).astype(np.uint8)
    assert np.array_equal(mc.cpu().numpy(), ref_q)
    # frame-at-a-time accumulation gives the same state (term indices drive the cascade, not call boundaries)
    st2 = engine.meanbg_new_state(72, 128)
    for f in range(F):
        engine.meanbg_accumulate(st2, clip["depth"][f], clip["rgb"][f], clip["mask"][f], f)
    assert torch.equal(st2, st)
    # mean cloud = unprojection of the mean maps with the ALL rule
    ref_p, ref_cols = O.build_mean_background_points(clip["depth"], clip["rgb"], clip["mask"], clip["intr"],
                                                     clip["c2w"], clip["T"], 2)
    cloud = engine.unproject_frame(md, mc, None, clip["intr"], clip["c2w"], clip["T"], subsample=2,
                                   select=_lib.SELECT_ALL, want_xyz64=True)
    n = cloud.n()
    assert n == ref_p.shape[0]
    assert np.array_equal(cloud.xyz64[:n].cpu().numpy(), ref_p)
    assert np.array_equal(cloud.to_numpy()[1], ref_cols)


@pytest.mark.parametrize("size", [(64, 126), (45, 77), (16, 8)])
def test_mean_background_ragged_sizes(engine, size):
    """Widths that are not a multiple of 4 take the one-pixel-per-thread kernel, small and odd sizes have partial
    workgroups on both axes; the oracle's plain left-to-right sums equal the cascade for fewer than 16 frames."""
    Hh, Ww = size
    F = 11
    clip = synthetic_clip(Hh, Ww, F)
    ref_d, ref_c = O.mean_background(clip["depth"], clip["rgb"], clip["mask"])
    st = engine.meanbg_new_state(Hh, Ww)
    engine.meanbg_accumulate(st, clip["depth"][:6], clip["rgb"][:6], clip["mask"][:6], 0)
    engine.meanbg_accumulate(st, clip["depth"][6:], clip["rgb"][6:], clip["mask"][6:], 6)
    md, mc = engine.meanbg_finalize(st)
    assert np.array_equal(md.cpu().numpy(), ref_d)
    assert np.array_equal(mc.cpu().numpy(), np.clip(ref_c * F32(255), 0, 255).astype(np.uint8))


def test_mean_background_matches_torch_nanmean(engine):
    """The same reduction through torch.nanmean on CPU (what the reference calls)."""
    clip = synthetic_clip(72, 128, 19)
    d = torch.from_numpy(clip["depth"].copy())
    rgb = torch.from_numpy(clip["rgb"]).float() / 255.0
    for f in range(19):
        ok = torch.from_numpy(O.reliable_depth_mask_range(clip["depth"][f]) & (clip["mask"][f] == 0))
        d[f][~ok] = float("nan")
        rgb[f][~ok] = float("nan")
    ref_d = torch.nan_to_num(torch.nanmean(d, dim=0), nan=0.0)
    ref_c = (torch.nan_to_num(torch.nanmean(rgb, dim=0), nan=0.0) * 255).clamp(0, 255).to(torch.uint8)
    st = engine.meanbg_new_state(72, 128)
    engine.meanbg_accumulate(st, clip["depth"], clip["rgb"], clip["mask"], 0)
    md, mc = engine.meanbg_finalize(st)
    assert torch.equal(md.cpu(), ref_d)
    assert torch.equal(mc.cpu(), ref_c)


@pytest.mark.parametrize("F", [256, 300, 1000])
def test_mean_background_long_clips(engine, F):
    """Clips long enough for the upper cascade levels: ATen flushes level 0 into level 1 every 16 terms, level 1 into
    level 2 every 256 and the tail is folded at the end; BASELINE configs[2] is 1 000 frames.  Checked against the
    oracle's cascade and against torch.nanmean itself (the call the reference makes), bit for bit; accumulated in calls
    of uneven length so that call boundaries fall inside runs."""
    clip = synthetic_clip(72, 128, F)
    ref_d, ref_c = O.mean_background(clip["depth"], clip["rgb"], clip["mask"])
    st = engine.meanbg_new_state(72, 128)
    cuts = [0, 7, 100, 256, 271, 513, F]
    cuts = sorted(set(c for c in cuts if c <= F))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        engine.meanbg_accumulate(st, clip["depth"][lo:hi], clip["rgb"][lo:hi], clip["mask"][lo:hi], lo)
    md, mc = engine.meanbg_finalize(st)
    assert np.array_equal(md.cpu().numpy(), ref_d)
    assert np.array_equal(mc.cpu().numpy(), np.clip(ref_c * F32(255), 0, 255).astype(np.uint8))
    st1 = engine.meanbg_new_state(72, 128)
    engine.meanbg_accumulate(st1, clip["depth"], clip["rgb"], clip["mask"], 0)  # one call for the whole clip
    assert torch.equal(st1, st)
    d = torch.from_numpy(clip["depth"].copy())
    rgb = torch.from_numpy(clip["rgb"]).float() / 255.0
    for f in range(F):
        ok = torch.from_numpy(O.reliable_depth_mask_range(clip["depth"][f]) & (clip["mask"][f] == 0))
        d[f][~ok] = float("nan")
        rgb[f][~ok] = float("nan")
    t_d = torch.nan_to_num(torch.nanmean(d, dim=0), nan=0.0)
    t_c = (torch.nan_to_num(torch.nanmean(rgb, dim=0), nan=0.0) * 255).clamp(0, 255).to(torch.uint8)
    assert torch.equal(md.cpu(), t_d)
    assert torch.equal(mc.cpu(), t_c)


def test_render_frame_applies_the_entry_filters(engine):
    """Clouds handed to egoprior_render_frame without a pass through egoprior_filter_cloud (per-frame dynamic points,
    say) still obey the reference's entry filters: rows with a non-finite coordinate or a float32 norm >= 1e6 are never
    drawn.  Same frame as filtering first, and as the oracle's."""
    from egoprior import engine as eng
    rng = np.random.default_rng(21)
    n = 4000
    pts = np.empty((n, 3), dtype=F32)
    pts[:, 0] = rng.uniform(0.8, 2.0, n)  # a cluster well to the right of the image centre
    pts[:, 1] = rng.uniform(-1.0, 1.0, n)
    pts[:, 2] = rng.uniform(2.0, 3.0, n)
    pts[5] = [0.0, 0.0, 1.0e6]          # norm == 1e6: dropped
    pts[6] = [3.0e3, 0.0, 999999.0]     # norm just above: dropped
    pts[7] = [5.0e2, 0.0, 9.99e5]       # norm below: drawn, next to the centre
    pts[8] = [np.nan, 0.0, 3.0]
    pts[9] = [0.0, np.inf, 3.0]
    pts[10] = [0.0, 0.0, -np.inf]
    pts[11] = [1e20, 1e20, 1e20]        # squares overflow to infinity
    cols = rng.integers(2, 256, (n, 3), dtype=np.uint8)
    He, We, ps = 64, 96, 4.0
    Kc = np.array([[0.8 * We, 0, We / 2], [0, 0.8 * We, He / 2], [0, 0, 1]], dtype=F32)
    T = np.eye(4, dtype=F32)
    ref = O.render_points_pinhole(pts, cols, Kc, T, We, He, ps)
    cam = eng.pinhole_camera(T[:3, :3], T[:3, 3], Kc, He, We)
    rp = eng.make_raster_params(ps, He, We, K=20, rotate_cw=True, colors_gt1=-1)
    good = engine.filter_cloud(pts, cols)
    assert good.n() == n - 6
    raw = engine.new_cloud(n)
    rec = np.zeros((n, 4), dtype=F32)
    rec[:, :3] = pts
    c = cols.astype(np.uint32)
    rec[:, 3] = (c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16)).view(F32)
    raw.points.copy_(torch.from_numpy(rec))
    raw.count.fill_(n)
    raw.flags.fill_(1)
    img_good = engine.render_frame(cam, rp, good)["image"].cpu().numpy()
    assert np.array_equal(img_good, ref)
    assert np.array_equal(engine.render_frame(cam, rp, raw)["image"].cpu().numpy(), ref)
    assert np.array_equal(engine.render_frame(cam, rp, good, raw)["image"].cpu().numpy(),
                          O.render_points_pinhole(np.vstack([pts, pts]), np.vstack([cols, cols]), Kc, T, We, He, ps))
    # the far point that passes the filter is really in the picture
    assert ref[We // 2 - 3:We // 2 + 4, He // 2 - 4:He // 2 + 4].any()


def test_bad_arguments_raise(engine):
    from egoprior import _lib, engine as eng
    rp = eng.make_raster_params(3.0, 32, 32)
    rp.K = 21
    ndc = torch.zeros((4, 4), device=engine.device)
    with pytest.raises(_lib.EgoPriorError):
        engine.rasterize_composite(ndc, None, rp, 32, 32)
    with pytest.raises(ValueError):
        eng.split_w2c(np.eye(3))
