"""The viewer's mean-background functions on the GPU kernels (vipe/utils/viser.py:60-74, 123-214).

Only the two numeric functions of the reference's viewer module live here -- the ones that duplicate the renderer's
mean-background code with another order of operations (frames are subsampled BEFORE they are stacked, the results stay
float32 torch tensors).  The interactive viewer itself (viser GUI) is not part of this package.
"""
from __future__ import annotations

import logging
from typing import Optional

from .io import ArtifactPath, read_depth_artifacts, read_instance_artifacts, read_rgb_artifacts

logger = logging.getLogger(__name__)


def compute_robust_mean_tensors(stacked_depth, stacked_rgb):
    """torch.nanmean over the frame axis, NaN -> 0 (vipe/utils/viser.py:60-74): the reference's own two torch calls, for
    callers that hold stacks already."""
    import torch
    return (torch.nan_to_num(torch.nanmean(stacked_depth, dim=0), nan=0.0),
            torch.nan_to_num(torch.nanmean(stacked_rgb, dim=0), nan=0.0))


def _none_after_missing(it):
    """The reference's none_it / none_it_mask (vipe/utils/viser.py:133-147)."""
    try:
        for item in it:
            yield item
    except FileNotFoundError:
        while True:
            yield None, None


def compute_mean_background_depth(artifact_path: ArtifactPath, spatial_subsample: int = 2,
                                  max_frames: Optional[int] = None):
    """(mean_depth (ceil(H/s), ceil(W/s)) float32, mean_rgb (ceil(H/s), ceil(W/s), 3) float32) as torch tensors on the
    GPU, or (None, None) when no frame has depth -- vipe/utils/viser.py:123-214.

    Per frame, at full resolution: keep pixels with instance id 0 (all pixels without a mask file) that pass the 5x5
    reliability test; then [::s, ::s]; then the NaN-aware mean over the kept frames, NaN -> 0.  Frames stream through the
    17-plane accumulator of the mean-background kernels instead of being stacked (the reference stacks 16 F H W / s^2
    bytes); a pixel's terms are added in frame order either way, so the bits are torch.nanmean's."""
    from . import render_vipe_pointcloud as rv
    E = rv._engine()
    state, n_used = None, 0
    for frame_idx, ((_, rgb), (_, depth), (_, mask)) in enumerate(zip(
            read_rgb_artifacts(artifact_path.rgb_path),
            _none_after_missing(read_depth_artifacts(artifact_path.depth_path, keep_half=True)),
            _none_after_missing(read_instance_artifacts(artifact_path.mask_path)))):
        if depth is None:
            continue
        if max_frames and n_used >= max_frames:
            break
        if state is None:
            state = E.meanbg_new_state(*depth.shape[:2])
        E.meanbg_accumulate(state, depth, rgb, mask, n_used)
        n_used += 1
        if frame_idx % 10 == 0:
            logger.info("Processed frame %d for mean background computation.", frame_idx)
    if state is None:
        logger.warning("No valid background frames found.")
        return None, None
    return E.meanbg_finalize_f32(state, int(spatial_subsample))
