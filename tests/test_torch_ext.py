"""The pybind11 / torch shim over the C ABI (csrc/egoprior_ext/), built in the convention of the reference's native modules
(csrc/bind.cpp:20-49: one sub-module per directory, torch::Tensor in and out, exceptions for errors).  On the CPU: the
module loads next to libegoprior_hip.so and refuses host tensors the way the reference's CHECK macros do.  On the GPU: the
same bits as the ctypes binding the rest of the package uses, on the default stream and on a side stream."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

CSRC = Path(__file__).resolve().parents[1] / "egox-egopriorrenderer_amd" / "csrc"


def _ext():
    if not (CSRC / "egoprior_torch_ext.so").exists():
        pytest.skip("csrc/egoprior_torch_ext.so is not built (make -C egox-egopriorrenderer_amd/csrc egoprior_torch_ext.so)")
    if str(CSRC) not in sys.path:
        sys.path.insert(0, str(CSRC))
    import egoprior_torch_ext
    return egoprior_torch_ext.egoprior_ext


def test_shim_loads_and_checks_its_inputs():
    m = _ext()
    for name in ("unproject_frame", "render_frame", "reliable_depth_mask"):
        assert callable(getattr(m, name))
    assert (m.SELECT_STATIC, m.SELECT_DYNAMIC, m.SELECT_ALL) == (0, 1, 2)
    assert (m.CAM_PINHOLE, m.CAM_FISHEYE) == (0, 1)
    with pytest.raises(RuntimeError, match="device tensor"):
        m.reliable_depth_mask(torch.zeros(8, 8))
    with pytest.raises(RuntimeError, match="device tensor"):
        m.unproject_frame(torch.zeros(8, 8), torch.zeros(8, 8, 3, dtype=torch.uint8), None, [1.0, 1.0, 4.0, 4.0],
                          torch.eye(4), torch.eye(4, dtype=torch.float64), 1, m.SELECT_STATIC)


def _camera_list(cam):
    v = list(cam.R) + list(cam.T) + [cam.fx, cam.fy, cam.px, cam.py]
    if cam.model == 1:
        v += list(cam.radial) + list(cam.tangential) + list(cam.thin_prism)
    return [float(x) for x in v]


@pytest.mark.gpu
@pytest.mark.parametrize("fisheye", [False, True])
def test_shim_gives_the_bits_of_the_ctypes_binding(engine, fisheye):
    from egoprior import _lib, engine as eng, synth
    m = _ext()
    H, W, He, We, ps = 120, 208, 96, 160, 3.0
    d, rgb, msk = synth.make_frames(H, W, [1], 4, device="cuda:0")
    exo = synth.exo_w2c()
    T = np.linalg.inv(exo)
    intr, c2w = synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32)
    ego = synth.ego_orbit_w2c(4, exo)[1]
    R, t = eng.split_w2c(ego)
    if fisheye:
        fx = synth.load_camera_fixture()
        prm = fx["online_calibration_line0"]["CameraCalibrations"][0]["Projection"]["Params"]
        from oracle import oracle as O
        f, cx, cy = O.scale_intrinsics_single(prm[0], prm[1], prm[2], (2880, 2880), (He, We))
        cam = eng.fisheye_camera(R, t, f, cx, cy, prm[3:9], prm[9:11], prm[11:15], He, We)
    else:
        cam = eng.pinhole_camera(R, t, synth.ego_pinhole_K(He, We), He, We)
    rp = eng.make_raster_params(ps, He, We, rotate_cw=not fisheye)
    # the package's own path
    bg = engine.unproject_frame(d[0], rgb[0], msk[0], intr, c2w, T, subsample=1, select=_lib.SELECT_STATIC)
    dyn = engine.unproject_frame(d[0], rgb[0], msk[0], intr, c2w, T, subsample=1, select=_lib.SELECT_DYNAMIC)
    ref = engine.render_frame(cam, rp, bg, dyn)
    assert ref["status"].item() == 0
    mask_ref = engine.reliable_depth_mask(d[0])
    torch.cuda.synchronize()

    def through_the_shim():
        c2w_t, T_t = torch.from_numpy(c2w), torch.from_numpy(T)
        a = m.unproject_frame(d[0], rgb[0], msk[0], [float(v) for v in intr], c2w_t, T_t, 1, m.SELECT_STATIC)
        b = m.unproject_frame(d[0].half(), rgb[0], msk[0], [float(v) for v in intr], c2w_t, T_t, 1, m.SELECT_DYNAMIC)
        img, status = m.render_frame(cam.model, He, We, _camera_list(cam), rp.radius, rp.radius_sq_w, rp.K,
                                     bool(rp.rotate_cw), rp.colors_gt1, a[0], a[1], a[2], b[0], b[1], b[2])
        return a, b, img, status, m.reliable_depth_mask(d[0])

    side = torch.cuda.Stream()
    for stream in (None, side):
        if stream is None:
            a, b, img, status, mk = through_the_shim()
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                a, b, img, status, mk = through_the_shim()
            stream.synchronize()
        torch.cuda.synchronize()
        assert status.item() == 0
        na, nb = int(a[1].item()), int(b[1].item())
        assert (na, nb) == (bg.n(), dyn.n()) and na > 0 and nb > 0
        assert torch.equal(a[0][:na].view(torch.int32), bg.points[:na].view(torch.int32))
        assert torch.equal(b[0][:nb].view(torch.int32), dyn.points[:nb].view(torch.int32))  # float16 depth: same bits
        assert int(a[2].item()) == int(bg.flags.item())
        assert torch.equal(img, ref["image"])
        assert torch.equal(mk.bool(), mask_ref)


@pytest.mark.gpu
def test_shim_turns_error_codes_into_exceptions(engine):
    m = _ext()
    pts = torch.zeros((16, 4), dtype=torch.float32, device="cuda:0")
    cnt = torch.full((1,), 16, dtype=torch.int32, device="cuda:0")
    flg = torch.zeros((1,), dtype=torch.int32, device="cuda:0")
    cam = [1.0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1.0, 1.0, 0.0, 0.0]
    with pytest.raises(RuntimeError, match="egoprior"):
        m.render_frame(0, 32, 32, cam, 0.05, 0.0025, 99, False, 1, pts, cnt, flg)  # K out of range
    with pytest.raises(RuntimeError, match="contiguous"):
        m.render_frame(0, 32, 32, cam, 0.05, 0.0025, 20, False, 1, pts.t().contiguous().t(), cnt, flg)
