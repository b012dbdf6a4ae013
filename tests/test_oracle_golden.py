"""CPU: the oracle against the committed golden vectors (outputs of the reference's own vipe/utils/cameras.py,
vipe/utils/depth.py and of torch.nanmean; see tests/golden/make_golden.py)."""
from pathlib import Path

import numpy as np
import torch

from oracle import oracle as O

G = Path(__file__).parent / "golden"


def test_ray_table_bit_exact():
    g = np.load(G / "g1_rays_mask.npz")
    for t in "abc":
        rays = O.pinhole_rays(24, 32, int(g[f"rays_{t}_s"]), g[f"rays_{t}_intr"])
        assert rays.dtype == np.float32 and np.array_equal(rays, g[f"rays_{t}"])


def test_reliability_mask_exact():
    g = np.load(G / "g1_rays_mask.npz")
    assert np.array_equal(O.reliable_depth_mask_range(g["depth"]), g["mask"])
    assert np.array_equal(O.reliable_depth_mask_range(g["depth2"]), g["mask2"])
    # corner of an all-ones image: zero padding in the mean, divisor always 25 -> ratio 0, still reliable
    assert O.reliable_depth_mask_range(np.ones((7, 9), np.float32)).all()


def test_nanmean_bit_exact_vs_golden():
    g = np.load(G / "g2_nanmean.npz")
    for F in (1, 5, 16, 17, 49, 300):
        assert np.array_equal(O.nanmean_stack(g[f"x_{F}"]), g[f"y_{F}"]), F


def test_nanmean_matches_torch_live():
    """Same check against torch itself (present on every box): shapes whose inner size is a multiple of 32 follow the
    cascade order the oracle restates."""
    rng = np.random.default_rng(7)
    for shape, F in (((16, 32), 33), ((40, 56, 3), 21), ((8, 64), 300), ((8, 32), 1000), ((4, 32), 4097)):
        x = rng.random((F,) + shape, dtype=np.float32)
        x[rng.random(x.shape) < 0.25] = np.nan
        ref = torch.nan_to_num(torch.nanmean(torch.from_numpy(x), dim=0), nan=0.0).numpy()
        assert np.array_equal(O.nanmean_stack(x), ref)


def test_u8_over_255_is_true_division():
    u = np.arange(256, dtype=np.uint8)
    ref = (torch.from_numpy(u) / 255.0).numpy()  # vipe/utils/io.py:250
    assert np.array_equal(u.astype(np.float32) / np.float32(255.0), ref)
    assert np.array_equal((ref * 255).astype(np.uint8), u)  # the colour round trip is the identity


def test_camera_fixture_values():
    from egoprior import synth
    fx = synth.load_camera_fixture()
    assert fx["take_name"] == "cmu_bike08_2"
    assert np.asarray(fx["cam01"]["camera_extrinsics"]).shape == (3, 4)
    assert len(fx["aria01"]["camera_extrinsics"]) == 128
    prm = fx["online_calibration_line0"]["CameraCalibrations"][0]["Projection"]["Params"]
    assert len(prm) == 15 and abs(prm[0] - 1220.592182729911) < 1e-9


def test_mei_and_panorama_ray_tables_match_the_reference_models():
    """G3: ray tables computed by the reference's own MeiCameraModel / PanoramaCameraModel (vipe/utils/cameras.py) with
    torch on CPU.  torch's vectorised float32 sqrt, sin and cos are not correctly rounded (they differ from libm in the
    last bit for ~0.5 % of arguments, and from CPU to CPU), and the Mei formula divides by (factor - k1), which
    amplifies that bit: the tables are compared to 2e-5 relative (Mei) and 2e-7 absolute (panorama)."""
    g = np.load(Path(__file__).parent / "golden" / "g3_rays_mei_pano.npz")
    H, W = int(g["H"]), int(g["W"])
    for tag in ("mei_a", "mei_b", "mei_c"):
        with np.errstate(invalid="ignore"):
            got = O.mei_rays(H, W, int(g[f"{tag}_s"]), g[f"{tag}_intr"])
        assert got.dtype == np.float32 and got.shape == g[tag].shape
        assert np.array_equal(np.isnan(got), np.isnan(g[tag])), tag  # k1 > 1: no ray beyond the model's field of view
        assert np.allclose(got, g[tag], rtol=2e-5, atol=0, equal_nan=True), tag
    assert np.array_equal(O.mei_rays(H, W, 3, g["mei_c_intr"]), g["mei_c"])  # k1 = 0 reduces to the pinhole table
    for tag in ("pano_a", "pano_b"):
        got = O.panorama_rays(H, W, int(g[f"{tag}_s"]))
        assert got.shape == g[tag].shape and np.abs(got - g[tag]).max() <= 2e-7, tag
