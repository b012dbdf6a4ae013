"""Full-size checks on the bench workload (1080p exo, ~2.1 M points, 480x832 ego, K = 20), where the CPU oracle would
take a minute per frame.  Instead of the oracle, these use properties that do not depend on the size:

* every splat kernel variant (independent implementations of the same selection) returns identical bits;
* fragments are ordered by (z, index) per pixel, every returned point really covers its pixel, and `dists` / `zbuf`
  are the point's own values;
* for a sample of pixels, a brute-force search over ALL points in plain torch float32 (separate mul / add kernels, so
  no fused multiply-add) finds exactly the same K indices in the same order;
* rendering twice gives the same bits (the tile lists are filled with atomics in a different order every time).
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

H, W, He, We = 1080, 1920, 480, 832
K = 20


@pytest.fixture(scope="module")
def scene(engine):
    from egoprior import _lib, synth
    from egoprior import engine as eng
    dev = engine.device
    d, rgb, m = synth.make_frames(H, W, [40], 300, device=dev)
    exo = synth.exo_w2c()
    T = np.linalg.inv(exo)
    intr, c2w = synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32)
    bg = engine.unproject_frame(d[0], rgb[0], m[0], intr, c2w, T, subsample=1, select=_lib.SELECT_STATIC)
    dyn = engine.unproject_frame(d[0], rgb[0], m[0], intr, c2w, T, subsample=1, select=_lib.SELECT_DYNAMIC)
    R, t = eng.split_w2c(synth.ego_orbit_w2c(300, exo)[40])
    cam = eng.pinhole_camera(R, t, synth.ego_pinhole_K(He, We), He, We)
    n = bg.n() + dyn.n()
    ndc = engine.project_points(cam, bg, dyn)[:n].clone()
    assert n > 1_500_000
    return dict(bg=bg, dyn=dyn, cam=cam, ndc=ndc, n=n)


def _render(engine, scene, ps, variant):
    from egoprior import _lib
    from egoprior import engine as eng
    rp = eng.make_raster_params(ps, He, We, K=K, rotate_cw=True, colors_gt1=-1)
    _lib.set_option("splat_variant", variant)
    try:
        res = engine.render_frame(scene["cam"], rp, scene["bg"], scene["dyn"], want_f32=True, want_fragments=True)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("splat_variant", 0)
    assert res["status"].item() == 0
    return res, rp


@pytest.mark.parametrize("ps", [5.0, 1.0])
def test_variants_agree_and_fragments_are_consistent(engine, scene, ps):
    ref, rp = _render(engine, scene, ps, 0)
    for variant in (1, 3, 0):  # plain insertion kernel, 1024-record chunks, and the default once more
        res, _ = _render(engine, scene, ps, variant)
        for key in ("idx", "zbuf", "dists", "image_f32", "image"):
            assert torch.equal(res[key], ref[key]), f"variant {variant}: {key} differs"
    idx, zb, ds = ref["idx"].long(), ref["zbuf"], ref["dists"]
    valid = idx >= 0
    assert valid.any() and (~valid).any()
    # filled slots form a prefix; unfilled slots hold -1 everywhere
    assert torch.equal(valid[..., 1:] & ~valid[..., :-1], torch.zeros_like(valid[..., 1:]))
    assert (zb[~valid] == -1).all() and (ds[~valid] == -1).all()
    ndc = scene["ndc"]
    z_of = ndc[:, 2][idx.clamp(min=0)]
    assert torch.equal(z_of[valid], zb[valid])
    # order: (z, index) strictly increasing along K
    both = valid[..., 1:] & valid[..., :-1]
    dz = zb[..., 1:] - zb[..., :-1]
    di = idx[..., 1:] - idx[..., :-1]
    assert ((dz > 0) | ((dz == 0) & (di > 0)))[both].all()
    # every returned point covers its pixel, and dists is its own squared distance (same float32 operations)
    xf, yf = O.pixel_ndc_centres(He, We)
    xf = torch.from_numpy(xf).to(ndc.device)[None, :, None]
    yf = torch.from_numpy(yf).to(ndc.device)[:, None, None]
    dx = ndc[:, 0][idx.clamp(min=0)] - xf
    dy = ndc[:, 1][idx.clamp(min=0)] - yf
    dx2 = dx * dx
    dy2 = dy * dy
    d2 = dx2 + dy2
    assert torch.equal(d2[valid], ds[valid])
    r2 = np.float32(rp.radius) * np.float32(rp.radius)
    assert (ds[valid] < float(r2)).all()


@pytest.mark.parametrize("ps", [5.0, 1.0])
def test_sampled_pixels_against_brute_force(engine, scene, ps):
    res, rp = _render(engine, scene, ps, 0)
    ndc = scene["ndc"]
    dev = ndc.device
    x, y, z = ndc[:, 0].contiguous(), ndc[:, 1].contiguous(), ndc[:, 2].contiguous()
    zbits = torch.where(z == 0, torch.zeros_like(z), z).view(torch.int32).long()
    key_all = (zbits << 32) | torch.arange(ndc.shape[0], device=dev)
    live = z >= 0  # also false for NaN
    xf, yf = O.pixel_ndc_centres(He, We)
    xf, yf = torch.from_numpy(xf).to(dev), torch.from_numpy(yf).to(dev)
    r2 = float(np.float32(rp.radius) * np.float32(rp.radius))
    rng = np.random.default_rng(7)
    pix = [(int(a), int(b)) for a, b in zip(rng.integers(0, He, 300), rng.integers(0, We, 300))]
    pix += [(yy, xx) for yy in (0, 7, 8, He - 1) for xx in (0, 7, 8, 415, 416, We - 1)]  # tile and image borders
    big = torch.iinfo(torch.int64).max
    for c0 in range(0, len(pix), 32):
        chunk = pix[c0:c0 + 32]
        py = torch.tensor([p[0] for p in chunk], device=dev)
        px = torch.tensor([p[1] for p in chunk], device=dev)
        dx = x[None, :] - xf[px][:, None]
        dy = y[None, :] - yf[py][:, None]
        dx2 = dx * dx
        dy2 = dy * dy
        d2 = dx2 + dy2
        hit = (d2 < r2) & live[None, :]
        keys = torch.where(hit, key_all[None, :], torch.full_like(key_all[None, :], big))
        best = torch.topk(keys, K, dim=1, largest=False, sorted=True).values
        want = torch.where(best == big, torch.full_like(best, -1), best & 0xFFFFFFFF).int()
        got = res["idx"][py, px]
        assert torch.equal(got, want), f"pixels {chunk}"


def test_two_runs_same_bits(engine, scene):
    a, _ = _render(engine, scene, 5.0, 0)
    keep = {k: a[k].clone() for k in ("idx", "zbuf", "dists", "image")}
    for _ in range(3):
        b, _ = _render(engine, scene, 5.0, 0)
        for k, v in keep.items():
            assert torch.equal(b[k], v)


@pytest.mark.parametrize("bin_lds", [1, 0])
def test_nine_million_points_against_the_oracle(engine, bin_lds):
    """More points than the binning grid has blocks for (4096 x 2048): its workgroups loop, and the 16-bit LDS counters
    of the counting pass take several blocks each.  Most discs miss the small image, so the point-driven oracle stays
    fast; idx / zbuf / dists must still be bit-exact."""
    from egoprior import _lib
    from egoprior import engine as eng
    rng = np.random.default_rng(3)
    n, Hs, Ws = 9_000_000, 64, 96
    ndc = np.empty((n, 3), dtype=np.float32)
    ndc[:, 0] = rng.uniform(-4, 4, n)
    ndc[:, 1] = rng.uniform(-3, 3, n)
    ndc[:, 2] = rng.uniform(0.5, 50, n)
    ndc[::1000, 2] = -1.0  # behind the camera
    r = O.radius_ndc(2.0, Hs, Ws)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, Hs, Ws, K, "scatter")
    rec = torch.zeros((n, 4), dtype=torch.float32, device=engine.device)
    rec[:, :3] = torch.from_numpy(ndc).to(engine.device)
    rp = eng.make_raster_params(2.0, Hs, Ws, K=K, colors_gt1=1)
    _lib.set_option("bin_lds", bin_lds)
    try:
        res = engine.rasterize_composite(rec, None, rp, Hs, Ws, want_fragments=True)
    finally:
        _lib.set_option("bin_lds", 1)
    assert res["status"].item() == 0
    assert np.array_equal(res["idx"].cpu().numpy(), ref_idx)
    assert np.array_equal(res["zbuf"].cpu().numpy(), ref_z)
    assert np.array_equal(res["dists"].cpu().numpy(), ref_d)


def test_too_many_point_tile_pairs_is_refused(engine):
    """List positions are 32-bit; a call whose discs could need more entries fails up front instead of overflowing."""
    import ctypes as C
    from egoprior import _lib
    from egoprior import engine as eng
    rp = eng.make_raster_params(40.0, 64, 64, K=K, colors_gt1=1)  # discs of 40 px: dozens of tiles each
    dummy = torch.zeros(64, dtype=torch.uint8, device=engine.device)
    img = torch.zeros((64, 64, 3), dtype=torch.uint8, device=engine.device)
    n = 0x7FFFFFF0  # nothing is dereferenced: the check comes before any launch
    rc = engine.L.egoprior_rasterize_composite(C.byref(rp), 64, 64, C.c_void_p(dummy.data_ptr()), n, None, None, None,
                                               C.c_void_p(img.data_ptr()), None, None, None, None, None,
                                               C.c_void_p(dummy.data_ptr()), dummy.numel(), None)
    assert rc == _lib.E_INVALID
    assert b"point-tile pairs" in engine.L.egoprior_last_error()


def test_twenty_million_points_on_a_1024_square_image(engine):
    """The scale of BASELINE configs[4] (tens of millions of accumulated points, 1024 x 1024 ego): about 1 900 points
    cover every pixel and every tile list is some twenty chunks long, so almost all of the work is in the later-chunk
    path (threshold filter, per-pixel prefix, merge network).  The point-driven oracle still finishes in seconds."""
    from egoprior import engine as eng
    rng = np.random.default_rng(5)
    n, S = 20_000_000, 1024
    ndc = np.empty((n, 3), dtype=np.float32)
    ndc[:, 0] = rng.uniform(-1.1, 1.1, n)
    ndc[:, 1] = rng.uniform(-1.1, 1.1, n)
    ndc[:, 2] = rng.uniform(0.5, 50, n)
    r = O.radius_ndc(3.0, S, S)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, S, S, K, "scatter")
    rec = torch.zeros((n, 4), dtype=torch.float32, device=engine.device)
    rec[:, :3] = torch.from_numpy(ndc).to(engine.device)
    rp = eng.make_raster_params(3.0, S, S, K=K, colors_gt1=1)
    res = engine.rasterize_composite(rec, None, rp, S, S, want_fragments=True)
    assert res["status"].item() == 0
    assert np.array_equal(res["idx"].cpu().numpy(), ref_idx)
    assert np.array_equal(res["zbuf"].cpu().numpy(), ref_z)
    assert np.array_equal(res["dists"].cpu().numpy(), ref_d)


def test_fiftyeight_million_points_on_a_1024_square_image(engine):
    """BASELINE configs[4] at its real size: ~58 M points (50 M accumulated + 8 M of a 4K exo frame) on a 1024 x 1024 ego
    image at point_size 5.  Every pixel is covered by ~3 500 discs and every tile list is dozens of chunks long.  9 tiles
    per disc at most, so the 32-bit list positions allow 477 M points at this radius; 58 M is what the config names."""
    from egoprior import engine as eng
    rng = np.random.default_rng(11)
    n, S = 58_000_000, 1024
    ndc = np.empty((n, 3), dtype=np.float32)
    ndc[:, 0] = rng.uniform(-1.1, 1.1, n)
    ndc[:, 1] = rng.uniform(-1.1, 1.1, n)
    ndc[:, 2] = rng.uniform(0.5, 50, n)
    ndc[::4099, 2] = -0.5  # behind the camera
    r = O.radius_ndc(5.0, S, S)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, S, S, K, "scatter")
    rec = torch.zeros((n, 4), dtype=torch.float32, device=engine.device)
    rec[:, :3] = torch.from_numpy(ndc).to(engine.device)
    del ndc
    rp = eng.make_raster_params(5.0, S, S, K=K, colors_gt1=1)
    res = engine.rasterize_composite(rec, None, rp, S, S, want_fragments=True)
    assert res["status"].item() == 0
    assert np.array_equal(res["idx"].cpu().numpy(), ref_idx)
    assert np.array_equal(res["zbuf"].cpu().numpy(), ref_z)
    assert np.array_equal(res["dists"].cpu().numpy(), ref_d)
    engine._ws.pop("raster", None)  # 8 GB of tile lists: give them back


@pytest.mark.parametrize("select", ["static", "dynamic"])
def test_unproject_4k_exo_frame(engine, select):
    """BASELINE configs[4]'s exo side: 3840 x 2160, subsample 1 = 8 100 compaction blocks, more than the scatter kernels
    sum on their own (4 096), so the one-workgroup block scan runs in between.  Bit-exact float64 points, colours, order."""
    from egoprior import _lib, synth
    Hh, Ww = 2160, 3840
    d, rgb, m = [t[0].numpy() for t in synth.make_frames(Hh, Ww, [3], 8)]
    exo = synth.exo_w2c()
    T = np.linalg.inv(exo)
    intr = synth.exo_intrinsics(Hh, Ww)
    c2w = np.eye(4, dtype=np.float32)
    a = 0.05
    c2w[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32)
    c2w[:3, 3] = [0.05, -0.1, 0.2]
    ref_p, ref_c = O.unproject_frame(d, rgb, m, intr, c2w, T, 1, select)
    sel = _lib.SELECT_STATIC if select == "static" else _lib.SELECT_DYNAMIC
    cloud = engine.unproject_frame(d, rgb, m, intr, c2w, T, subsample=1, select=sel, want_xyz64=True)
    n = cloud.n()
    assert n == ref_p.shape[0] and n > (1_000_000 if select == "static" else 10_000)
    assert np.array_equal(cloud.xyz64[:n].cpu().numpy(), ref_p)
    p32, cols = cloud.to_numpy()
    assert np.array_equal(cols, ref_c)
    assert np.array_equal(p32, ref_p.astype(np.float32))
    assert cloud.flags.item() == 1
    engine._ws.pop("unproject", None)


def test_exact_workspace_sizing_for_large_discs(engine):
    """point_size 20 on a 1024 x 1024 image: a disc can overlap 49 tiles, so the worst-case workspace of 20 M points is
    7.8 GB although most points lie outside the image.  The two-call mode counts first (egoprior_count_list_entries), and
    a workspace sized from that count renders the same bits as the oracle; one entry less and the status word reports
    the overflow."""
    import ctypes as C
    from egoprior import _lib
    from egoprior import engine as eng
    L, dev = engine.L, engine.device
    rng = np.random.default_rng(17)
    n, S, ps = 20_000_000, 1024, 20.0
    ndc = np.empty((n, 3), dtype=np.float32)
    ndc[:, 0] = rng.uniform(-3.0, 3.0, n)
    ndc[:, 1] = rng.uniform(-3.0, 3.0, n)
    ndc[:, 2] = rng.uniform(0.5, 50, n)
    r = O.radius_ndc(ps, S, S)
    ref_idx, ref_z, ref_d = O.rasterize(ndc, r, S, S, K, "scatter")
    rec = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    rec[:, :3] = torch.from_numpy(ndc).to(dev)
    del ndc
    rp = eng.make_raster_params(ps, S, S, K=K, colors_gt1=1)
    worst = L.egoprior_raster_workspace_bytes(S, S, n, rp.radius)
    ws0 = torch.empty(L.egoprior_raster_workspace_bytes_entries(S, S, n, 0), dtype=torch.uint8, device=dev)
    entries = torch.zeros(1, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.egoprior_count_list_entries(None, C.byref(rp), S, S, C.c_void_p(rec.data_ptr()), n, None, None, 0, None,
                                             C.c_void_p(entries.data_ptr()), C.c_void_p(ws0.data_ptr()), ws0.numel(), st))
    need = int(entries.item())
    exact = L.egoprior_raster_workspace_bytes_entries(S, S, n, need)
    assert 0 < need < n * 49 and exact < worst // 4
    # every tile list entry is a (disc, tile) pair whose rectangle the binning accepted: the count is at least the
    # number of hits divided by the pixels of a tile
    assert need >= int((ref_idx >= 0).any(axis=2).sum()) // 64

    def render(nbytes):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        idx = torch.empty((S, S, K), dtype=torch.int32, device=dev)
        zb = torch.empty((S, S, K), dtype=torch.float32, device=dev)
        ds = torch.empty((S, S, K), dtype=torch.float32, device=dev)
        img = torch.empty((S, S, 3), dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.egoprior_rasterize_composite(C.byref(rp), S, S, C.c_void_p(rec.data_ptr()), n, None, None, None,
                                                  C.c_void_p(img.data_ptr()), None, C.c_void_p(idx.data_ptr()),
                                                  C.c_void_p(zb.data_ptr()), C.c_void_p(ds.data_ptr()),
                                                  C.c_void_p(status.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), st))
        torch.cuda.synchronize()
        return idx, zb, ds, int(status.item())

    idx, zb, ds, status = render(exact)
    assert status == 0
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(zb.cpu().numpy(), ref_z)
    assert np.array_equal(ds.cpu().numpy(), ref_d)
    _, _, _, status = render(L.egoprior_raster_workspace_bytes_entries(S, S, n, need - 300))
    assert status == 1  # the lists no longer fit: reported, nothing written out of bounds


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at its real size, and the command-line tool's default workload
# ----------------------------------------------------------------------------------------------------------------------
def _fisheye_cam(frame_no, n_frames, S):
    from egoprior import synth
    from egoprior import engine as eng
    from egoprior.render_vipe_pointcloud import load_aria_distortion_coeffs
    radial, tangential, thin_prism, _, _ = load_aria_distortion_coeffs(None)
    exo = synth.exo_w2c()
    R, t = eng.split_w2c(synth.ego_orbit_w2c(n_frames, exo)[frame_no])
    K = synth.ego_pinhole_K(S, S)  # the ego intrinsics bench.py --fisheye passes: f = 0.6 S, centred
    return eng.fisheye_camera(R, t, K[0, 0], K[0, 2], K[1, 2], radial, tangential, thin_prism, S, S)


def _brute_force_pixels(ndc, res, rp, S, pix):
    """Plain torch float32 search over ALL points for the given pixels: the K smallest (z, index) among the discs that
    cover the pixel centre, compared with the kernel's index map."""
    dev = ndc.device
    x, y, z = ndc[:, 0].contiguous(), ndc[:, 1].contiguous(), ndc[:, 2].contiguous()
    zbits = torch.where(z == 0, torch.zeros_like(z), z).view(torch.int32).long()
    key_all = (zbits << 32) | torch.arange(ndc.shape[0], device=dev)
    live = z >= 0
    xf, yf = O.pixel_ndc_centres(S, S)
    xf, yf = torch.from_numpy(xf).to(dev), torch.from_numpy(yf).to(dev)
    r2 = float(np.float32(rp.radius) * np.float32(rp.radius))
    big = torch.iinfo(torch.int64).max
    for c0 in range(0, len(pix), 16):
        chunk = pix[c0:c0 + 16]
        py = torch.tensor([p[0] for p in chunk], device=dev)
        px = torch.tensor([p[1] for p in chunk], device=dev)
        dx = x[None, :] - xf[px][:, None]
        dy = y[None, :] - yf[py][:, None]
        dx2 = dx * dx
        dy2 = dy * dy
        d2 = dx2 + dy2
        hit = (d2 < r2) & live[None, :]
        keys = torch.where(hit, key_all[None, :], torch.full_like(key_all[None, :], big))
        best = torch.topk(keys, K, dim=1, largest=False, sorted=True).values
        want = torch.where(best == big, torch.full_like(best, -1), best & 0xFFFFFFFF).int()
        assert torch.equal(res["idx"][py, px], want), f"pixels {chunk}"


def test_fisheye_448_point_size_20_at_full_size(engine, scene):
    """BASELINE configs[3] as the reference runs it (scripts/render_vipe.sh: fisheye, 448 x 448) with the static and the
    dynamic points of a 1080p exo frame and point_size 20: discs of 20 px radius, tens of tiles per disc, and -- the
    fisheye periphery packs far more points into some tiles than others -- tile lists long enough to be cut into slices.
    Variants agree, slicing on and off agrees, sampled pixels match a brute-force search over all points."""
    from egoprior import _lib
    from egoprior import engine as eng
    S, ps = 448, 20.0
    cam = _fisheye_cam(40, 300, S)
    rp = eng.make_raster_params(ps, S, S, K=K, rotate_cw=False, colors_gt1=-1)

    def render(variant, slices):
        _lib.set_option("splat_variant", variant)
        _lib.set_option("slices", slices)
        try:
            r = engine.render_frame(cam, rp, scene["bg"], scene["dyn"], want_f32=True, want_fragments=True)
            torch.cuda.synchronize()
        finally:
            _lib.set_option("splat_variant", 0)
            _lib.set_option("slices", 1)
        assert r["status"].item() == 0
        return r

    ref = render(0, 1)
    covered = (ref["idx"][..., 0] >= 0).float().mean().item()
    assert covered > 0.25, covered
    assert (ref["idx"][..., K - 1] >= 0).float().mean().item() > 0.8 * covered  # covered pixels hold all K fragments
    for variant, slices in ((0, 0), (3, 1), (1, 1)):
        got = render(variant, slices)
        for key in ("idx", "zbuf", "dists", "image_f32", "image"):
            assert torch.equal(got[key], ref[key]), f"variant {variant}, slices {slices}: {key} differs"
    n = scene["n"]
    ndc = engine.project_points(cam, scene["bg"], scene["dyn"])[:n]
    rng = np.random.default_rng(21)
    pix = [(int(a), int(b)) for a, b in zip(rng.integers(0, S, 96), rng.integers(0, S, 96))]
    pix += [(yy, xx) for yy in (0, 7, 8, 223, 224, S - 1) for xx in (0, 7, 8, 223, 224, S - 1)]
    _brute_force_pixels(ndc, ref, rp, S, pix)
    engine._ws.pop("render", None)


def test_accumulated_cloud_of_the_cli_default_mode(engine, check_build):
    """What `python scripts/render_vipe_pointcloud.py` renders without --use_mean_bg: the static points of every frame of
    a clip accumulated into one cloud (scripts/render_vipe_pointcloud.py:304-456) -- 64 frames of 1080p at subsample 2,
    more than 30 M points, 64 near-copies of every surface point -- on the 448 x 448 fisheye image at point_size 5.  A few
    hundred tiles hold hundreds of thousands of entries each and are cut into up to 64 slices.  Index maps, z-buffers and
    distances against the oracle's scatter rasterizer, with slicing on and off; then the same frame through the -DEP_CHECK
    build, in which every slice stores a checksum behind its per-pixel lists and the merging wave verifies that what it
    read is what was written."""
    import ctypes as C
    from egoprior import _lib, synth
    from egoprior import engine as eng
    dev = engine.device
    F, Hx, Wx, S, ps = 64, 1080, 1920, 448, 5.0
    exo = synth.exo_w2c()
    T = np.linalg.inv(exo)
    intr, c2w = synth.exo_intrinsics(Hx, Wx), np.eye(4, dtype=np.float32)
    ns = (Hx // 2) * (Wx // 2)
    big = torch.empty((F * ns, 4), dtype=torch.float32, device=dev)
    tmp = engine.new_cloud(ns)
    total = 0
    for f0 in range(0, F, 16):
        d, rgb, m = synth.make_frames(Hx, Wx, list(range(f0, f0 + 16)), F, device=dev)
        for j in range(16):
            engine.unproject_frame(d[j], rgb[j], m[j], intr, c2w, T, subsample=2, select=_lib.SELECT_STATIC, out=tmp)
            n = tmp.n()
            big[total:total + n] = tmp.points[:n]
            total += n
        del d, rgb, m
    assert total > 30_000_000
    cloud = eng.DeviceCloud(points=big[:total], count=torch.tensor([total], dtype=torch.int32, device=dev),
                            flags=torch.ones(1, dtype=torch.int32, device=dev))
    cam = _fisheye_cam(10, F, S)
    ndc = engine.project_points(cam, cloud)[:total].contiguous()
    del big, cloud
    rp = eng.make_raster_params(ps, S, S, K=K, colors_gt1=1)
    ref_idx, ref_z, ref_d = O.rasterize(ndc[:, :3].cpu().numpy(), O.radius_ndc(ps, S, S), S, S, K, "scatter")
    assert (ref_idx[..., K - 1] >= 0).mean() > 0.2

    def check(E, slices):
        E.L.egoprior_set_option(b"slices", slices)
        try:
            res = E.rasterize_composite(ndc, None, rp, S, S, want_fragments=True)
            torch.cuda.synchronize()
        finally:
            E.L.egoprior_set_option(b"slices", 1)
        assert res["status"].item() == 0
        assert np.array_equal(res["idx"].cpu().numpy(), ref_idx), f"slices {slices}: idx"
        assert np.array_equal(res["zbuf"].cpu().numpy(), ref_z), f"slices {slices}: zbuf"
        assert np.array_equal(res["dists"].cpu().numpy(), ref_d), f"slices {slices}: dists"

    check(engine, 1)
    check(engine, 0)
    engine._ws.pop("raster", None)
    E2, L2 = check_build
    E2.check_counts()
    launches = int(os.environ.get("EGOPRIOR_HANDOVER_LAUNCHES", "3"))  # more for a soak run
    for _ in range(launches):  # the slot memory is reused from launch to launch: stale lists of the launch before must not pass
        check(E2, 1)
    seen, failed = E2.check_counts()
    print(f"slice hand-overs verified: {seen}, checksum mismatches: {failed}")
    assert seen > launches * 1000, seen   # a few hundred sliced tiles with dozens of slices each, per launch
    assert failed == 0
