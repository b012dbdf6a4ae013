"""GPU: the viewer's mean background (vipe/utils/viser.py:123-214) on the mean-background kernels, against the reference's
own recipe run in torch on the CPU: mask at full resolution (5x5 reliability test and instance id 0), [::s, ::s] BEFORE
stacking, torch.nanmean over the stack, NaN -> 0.  Bit-exact; the subsampled frames keep a multiple of 32 elements, the
condition under which ATen's reduction adds a pixel's terms in frame order."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _reference_recipe(depth, rgb_u8, mask, s, max_frames=None):
    """compute_mean_background_depth as the reference writes it, on CPU tensors (the reliability mask from the oracle's
    restatement of reliable_depth_mask_range, which the golden fixture pins to the reference's own function)."""
    dd, cc = [], []
    for f in range(depth.shape[0]):
        if max_frames and len(dd) >= max_frames:
            break
        d = torch.from_numpy(depth[f].astype(np.float32))
        c = torch.from_numpy(rgb_u8[f].astype(np.float32) / np.float32(255.0))
        keep = torch.from_numpy(O.reliable_depth_mask_range(depth[f].astype(np.float32)))
        if mask is not None:
            keep = keep & torch.from_numpy(mask[f] == 0)
        d = d.clone()
        c = c.clone()
        d[~keep] = float("nan")
        c[~keep.unsqueeze(-1).expand_as(c)] = float("nan")
        if s > 1:
            d, c = d[::s, ::s], c[::s, ::s]
        dd.append(d)
        cc.append(c)
    sd, sc = torch.stack(dd, 0), torch.stack(cc, 0)
    return (torch.nan_to_num(torch.nanmean(sd, dim=0), nan=0.0), torch.nan_to_num(torch.nanmean(sc, dim=0), nan=0.0))


@pytest.fixture(scope="module")
def clip(tmp_path_factory):
    from egoprior import io as eio, synth
    F, H, W = 40, 128, 192  # more than two runs of 16 frames; 64 x 96 and 32 x 48 after subsampling
    d, rgb, m = [t.numpy() for t in synth.make_frames(H, W, range(F), F)]
    d = d.copy()
    d[3, 10:20, 30:50] = np.nan      # holes, and one pixel block that no frame keeps
    d[:, 100:104, 8:12] = 0.0
    base = tmp_path_factory.mktemp("viser") / "take"
    art = eio.write_synthetic_artifacts(base, "exo", d, rgb, m, synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32),
                                        depth_as_exr_zip=True)
    base2 = tmp_path_factory.mktemp("viser_nomask") / "take"
    art2 = eio.write_synthetic_artifacts(base2, "exo", d, rgb, None, synth.exo_intrinsics(H, W),
                                         np.eye(4, dtype=np.float32))
    return dict(art=art, art_nomask=art2, depth=d.astype(np.float16), rgb=rgb, mask=m)


@pytest.mark.parametrize("s,max_frames", [(2, None), (1, None), (4, None), (2, 17), (2, 40), (2, 1000)])
def test_viewer_mean_background_matches_torch_nanmean(engine, clip, s, max_frames):
    from egoprior import viser as ev
    md, mc = ev.compute_mean_background_depth(clip["art"], spatial_subsample=s, max_frames=max_frames)
    rd, rc = _reference_recipe(clip["depth"], clip["rgb"], clip["mask"], s, max_frames)
    assert isinstance(md, torch.Tensor) and md.is_cuda and md.dtype == torch.float32 and mc.dtype == torch.float32
    assert tuple(md.shape) == tuple(rd.shape) and tuple(mc.shape) == tuple(rc.shape)
    assert np.array_equal(md.cpu().numpy().view(np.uint32), rd.numpy().view(np.uint32))
    assert np.array_equal(mc.cpu().numpy().view(np.uint32), rc.numpy().view(np.uint32))
    assert (rd == 0).any() and (rd > 0).any()  # the block no frame keeps became 0, not NaN


def test_viewer_mean_background_without_masks_and_without_depth(engine, clip, tmp_path):
    from egoprior import io as eio, viser as ev
    md, mc = ev.compute_mean_background_depth(clip["art_nomask"], spatial_subsample=2)
    rd, rc = _reference_recipe(clip["depth"], clip["rgb"], None, 2)
    assert np.array_equal(md.cpu().numpy().view(np.uint32), rd.numpy().view(np.uint32))
    assert np.array_equal(mc.cpu().numpy().view(np.uint32), rc.numpy().view(np.uint32))
    # no depth archive at all: (None, None) like the reference (vipe/utils/viser.py:201-203)
    base = tmp_path / "take"
    (base / "rgb").mkdir(parents=True)
    np.save(base / "rgb" / "exo.npy", clip["rgb"][:3])
    assert ev.compute_mean_background_depth(eio.ArtifactPath(base, "exo")) == (None, None)
