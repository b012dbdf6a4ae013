"""CPU: the host half of the viewer's mean-background functions (egoprior/viser.py <- vipe/utils/viser.py:60-74, 123-214)."""
import inspect
from pathlib import Path

import numpy as np
import torch

G = Path(__file__).parent / "golden"


def test_robust_mean_tensors_reproduces_the_golden_nanmean():
    """compute_robust_mean_tensors is the reference's two torch calls; g2_nanmean.npz holds torch.nanmean's outputs."""
    from egoprior import viser as ev
    g = np.load(G / "g2_nanmean.npz")
    for F in (1, 5, 16, 17, 49, 300):
        x = torch.from_numpy(g[f"x_{F}"])
        d, c = ev.compute_robust_mean_tensors(x, x[..., None].expand(*x.shape, 3).contiguous())
        assert np.array_equal(d.numpy(), g[f"y_{F}"])
        assert np.array_equal(c[..., 1].numpy(), g[f"y_{F}"])


def test_signature_is_the_viewers():
    from egoprior import viser as ev
    sig = inspect.signature(ev.compute_mean_background_depth)
    assert list(sig.parameters) == ["artifact_path", "spatial_subsample", "max_frames"]
    assert sig.parameters["spatial_subsample"].default == 2 and sig.parameters["max_frames"].default is None
    assert list(inspect.signature(ev.compute_robust_mean_tensors).parameters) == ["stacked_depth", "stacked_rgb"]


def test_no_gpu_no_result(tmp_path):
    """There is no CPU path: without a GPU the function raises instead of silently computing on the host."""
    import pytest
    from egoprior import io as eio, synth, viser as ev
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d, rgb, m = [t.numpy() for t in synth.make_frames(32, 64, range(2), 2)]
    art = eio.write_synthetic_artifacts(tmp_path / "take", "exo", d, rgb, m, synth.exo_intrinsics(32, 64),
                                        np.eye(4, dtype=np.float32))
    with pytest.raises(RuntimeError):
        ev.compute_mean_background_depth(art)
