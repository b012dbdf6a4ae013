"""A whole ViPE clip decoded once and kept in HBM.

The reference walks the artifact iterators again for every rendered frame (scripts/render_vipe_pointcloud.py:682-692) and
once more for the background cloud (:331-339, :500-509).  On an MI355X the renderer itself needs a fraction of a
millisecond per frame, so the command-line tool is bound by how the exo frames reach the GPU.  DeviceClip decodes every
frame exactly once -- zip members are read and decoded by worker threads (the native decoders release the GIL) straight into
page-locked staging buffers, which a copy stream moves into three resident tensors -- and then hands the frames out, as
device tensors, to both the cloud builder and the frame loop.  A 1080p frame is 12.4 MB (float16 depth, RGB, mask): a
thousand-frame clip is 12.4 GB of the 288 GB of HBM.

Depth (zip of EXR files or the .npy twin) and masks (zip of PNG files or the .npy twin) are read frame by frame in any
order by the worker threads; RGB comes from the .npy twin the same way, or from the mp4, which ffmpeg has to decode in
order: one more thread walks the video reader and uploads frame after frame while the workers decode depth and masks.
load() returns None for what is left (EXR files when the OpenEXR library is importable and therefore in charge; per-frame
depth arrays) and the caller streams the clip through egoprior.io's iterators instead.
"""
from __future__ import annotations

import os
import queue
import threading
import zipfile
from concurrent.futures import ThreadPoolExecutor, as_completed
from pathlib import Path
from typing import Iterator, Optional

import numpy as np
import torch

from . import io as _io


class _ZipMembers:
    """The members of a zip archive in sorted-name order (the order the reference's readers use, vipe/utils/io.py:288,
    :321), readable from several threads: every thread opens the archive for itself."""

    def __init__(self, path: Path):
        self.path = Path(path)
        with zipfile.ZipFile(self.path, "r") as z:
            self.names = sorted(z.namelist())
        self._tls = threading.local()
        self._all = []
        self._lock = threading.Lock()

    def __len__(self):
        return len(self.names)

    def read(self, i: int) -> bytes:
        z = getattr(self._tls, "z", None)
        if z is None:
            z = self._tls.z = zipfile.ZipFile(self.path, "r")
            with self._lock:
                self._all.append(z)
        return z.read(self.names[i])

    def close(self):
        for z in self._all:
            z.close()
        self._all.clear()


class _Source:
    """One stream of per-frame images: n frames of `shape` and `dtype`, and fill(i, out) that decodes frame i into the
    C-contiguous host array `out`."""
    sequential = False
    n: int
    shape: tuple
    dtype: np.dtype

    def fill(self, i: int, out: np.ndarray) -> None:
        raise NotImplementedError

    def close(self):
        pass


class _ArraySource(_Source):
    def __init__(self, path: Path, dtype=None):
        self.arr = np.load(path, mmap_mode="r")
        self.n, self.shape = self.arr.shape[0], tuple(self.arr.shape[1:])
        self.dtype = np.dtype(dtype) if dtype is not None else self.arr.dtype

    def fill(self, i, out):
        np.copyto(out, self.arr[i], casting="unsafe")


class _VideoSource(_Source):
    """rgb/<name>.mp4 through egoprior.io.read_rgb_artifacts (imageio + ffmpeg; without imageio that reader takes the
    .npy twin): frames arrive in order only.  n is unknown until the stream ends."""
    sequential = True
    dtype = np.dtype(np.uint8)

    def __init__(self, path: Path):
        self.path = Path(path)
        self._it = _io.read_rgb_artifacts(self.path)
        self._first = next(self._it, None)
        self.n = None
        self.shape = tuple(self._first[1].shape) if self._first is not None else (0, 0, 3)

    def frames(self):
        """(index, frame) in order, each frame once."""
        if self._first is not None:
            yield self._first
            self._first = None
        yield from self._it

    def close(self):
        close = getattr(self._it, "close", None)
        if close is not None:
            close()


class _ExrZipSource(_Source):
    """depth/<name>.zip: %05d.exr files with one HALF channel Z (vipe/utils/io.py:268-279).  A member that cannot be read
    becomes an all-NaN frame, as in the reference's reader (:293-305)."""
    dtype = np.dtype(np.float16)

    def __init__(self, path: Path):
        self.z = _ZipMembers(path)
        self.n = len(self.z)
        self.shape = _io.exr_size(self.z.read(0)) if self.n else (0, 0)

    def fill(self, i, out):
        try:
            _io.decode_exr_half_z(self.z.read(i), out=out)
        except OSError:
            out[...] = np.nan

    def close(self):
        self.z.close()


class _PngZipSource(_Source):
    """mask/<name>.zip: %05d.png files, 8-bit greyscale instance ids (vipe/utils/io.py:353-370)."""
    dtype = np.dtype(np.uint8)

    def __init__(self, path: Path):
        self.z = _ZipMembers(path)
        self.n = len(self.z)
        self.shape = tuple(_io.decode_instance_png(self.z.read(0)).shape) if self.n else (0, 0)

    def fill(self, i, out):
        data = self.z.read(i)
        if _io._native_decode("png_gray8", data, np.uint8, out) is None:
            out[...] = _io.decode_instance_png(data)  # not an 8-bit greyscale file, or no native decoder

    def close(self):
        self.z.close()


def _depth_source(artifact) -> Optional[_Source]:
    p = Path(artifact.depth_path)
    if p.exists():
        try:
            import OpenEXR  # noqa: F401  the library reads every EXR flavour; leave such clips to the streaming reader
            return None
        except ImportError:
            pass
        return _ExrZipSource(p)
    tw = p.with_suffix(".npy")
    if tw.exists():
        a = np.load(tw, mmap_mode="r")
        return _ArraySource(tw, np.float16 if a.dtype == np.float16 else np.float32)
    if p.with_suffix("").is_dir():
        return None  # per-frame .npy files: streamed
    raise FileNotFoundError(str(p))


def _mask_source(artifact) -> Optional[_Source]:
    p = Path(artifact.mask_path)
    if p.exists():
        return _PngZipSource(p)
    tw = p.with_suffix(".npy")
    if tw.exists():
        return _ArraySource(tw, np.uint8)
    raise FileNotFoundError(str(p))


class DeviceClip:
    """Frames of one artifact in HBM.  depth: (n, H, W) float16 / float32 or None (no depth container), rgb: (n, H, W, 3)
    uint8, mask: (n, H, W) uint8 or None (no mask container); position i of the tensors is frame index[i] of the clip."""

    #: clips whose resident size exceeds this many bytes are streamed instead (a thousand 1080p frames are 12.4e9)
    BUDGET_BYTES = 96 << 30

    def __init__(self):
        self.poses = self.intr = self.types = None
        self.depth = self.rgb = self.mask = None
        self.depth_missing = self.mask_missing = False
        self.n_frames = 0
        self.index = []     # frame numbers held, ascending
        self._pos = {}      # frame number -> position in the tensors

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.depth, self.rgb, self.mask) if t is not None)

    def has(self, f: int) -> bool:
        return f in self._pos

    def frame(self, f: int):
        """(c2w, rgb, intr, camera type, depth, mask) of frame f like one item of the reference's zip(...) over the
        artifact readers (:682-692); device tensors for the three images, None for a container that does not exist."""
        k = self._pos[f]
        return (self.poses[f], self.rgb[k], self.intr[f], self.types[f],
                None if self.depth is None else self.depth[k], None if self.mask is None else self.mask[k])

    def frames(self, tolerant: bool) -> Iterator[tuple]:
        """All frames in order.  tolerant=False raises FileNotFoundError for a missing depth / mask container, as the
        reference's frame loop does (:675); tolerant=True yields None in its place (the cloud builders' none_it)."""
        if not tolerant and (self.depth_missing or self.mask_missing):
            raise FileNotFoundError("depth or mask container of the clip is missing")
        for f in range(self.n_frames):
            yield self.frame(f) if f in self._pos else (self.poses[f], _io.SKIPPED, self.intr[f], self.types[f],
                                                        _io.SKIPPED, _io.SKIPPED)

    # ------------------------------------------------------------------------------------------------------------------
    @classmethod
    def load(cls, engine, artifact, wanted=None, workers: Optional[int] = None) -> Optional["DeviceClip"]:
        """Decodes the clip into HBM; None when a container cannot be read frame by frame or the clip is too large.
        wanted: a container of frame numbers -- only those are decoded and held (a rank of a multi-GPU run)."""
        clip = cls()
        _, clip.poses = _io.read_pose_artifacts(artifact.pose_path)
        _, clip.intr, clip.types = _io.read_intrinsics_artifacts(artifact.intrinsics_path, artifact.camera_type_path)
        rgb_path = Path(artifact.rgb_path)
        tw = rgb_path if rgb_path.suffix == ".npy" else rgb_path.with_suffix(".npy")
        if rgb_path.exists() and rgb_path.suffix != ".npy":
            rgb_src = _VideoSource(rgb_path)
        elif tw.exists():
            rgb_src = _ArraySource(tw, np.uint8)
        else:
            raise FileNotFoundError(str(rgb_path))
        srcs = [rgb_src]
        try:
            try:
                depth_src = _depth_source(artifact)
                if depth_src is None:
                    return None
                srcs.append(depth_src)
            except FileNotFoundError:
                depth_src, clip.depth_missing = None, True
            try:
                mask_src = _mask_source(artifact)
                srcs.append(mask_src)
            except FileNotFoundError:
                mask_src, clip.mask_missing = None, True
            # the reference zips the readers: the shortest one ends the clip
            n = min([len(clip.poses), len(clip.intr), len(clip.types)] + [s.n for s in srcs if s.n is not None])
            clip.n_frames = n
            if n == 0:
                return clip
            H, W = rgb_src.shape[:2]
            for s in srcs[1:]:
                if tuple(s.shape) != (H, W):
                    return None  # containers of different frame sizes: let the streaming path report it
            clip.index = [f for f in range(n) if wanted is None or f in wanted]
            clip._pos = {f: k for k, f in enumerate(clip.index)}
            m = len(clip.index)
            per_frame = H * W * (3 + (depth_src.dtype.itemsize if depth_src else 0) + (1 if mask_src else 0))
            if m * per_frame > cls.BUDGET_BYTES:
                return None
            dev = engine.device
            try:
                clip.rgb = torch.empty((m, H, W, 3), dtype=torch.uint8, device=dev)
                if depth_src is not None:
                    clip.depth = torch.empty((m, H, W), device=dev,
                                             dtype=torch.float16 if depth_src.dtype == np.float16 else torch.float32)
                if mask_src is not None:
                    clip.mask = torch.empty((m, H, W), dtype=torch.uint8, device=dev)
            except torch.cuda.OutOfMemoryError:
                return None  # the card is shared and too full: stream the clip instead
            if m:
                got = _fill(clip, rgb_src, depth_src, mask_src, workers)
                if got < n:  # the video ended before the other readers: it ends the clip, as in the reference's zip(...)
                    clip.n_frames = got
                    clip.index = [f for f in clip.index if f < got]
                    clip._pos = {f: k for k, f in enumerate(clip.index)}
            return clip
        finally:
            for s in srcs:
                s.close()


class _Staging:
    """Page-locked host buffers for one frame in flight.  Kept for the life of the process, per frame geometry: pinning
    memory costs far more than decoding a frame."""
    _pool = {}
    _lock = threading.Lock()

    def __init__(self, H, W, depth_dtype, with_mask):
        self.rgb = torch.empty((H, W, 3), dtype=torch.uint8, pin_memory=True)
        self.depth = None if depth_dtype is None else torch.empty((H, W), dtype=depth_dtype, pin_memory=True)
        self.mask = torch.empty((H, W), dtype=torch.uint8, pin_memory=True) if with_mask else None
        self.rgb_np = self.rgb.numpy()
        self.depth_np = None if self.depth is None else self.depth.numpy()
        self.mask_np = None if self.mask is None else self.mask.numpy()

    @classmethod
    def take(cls, key, count):
        with cls._lock:
            have = cls._pool.setdefault(key, [])
            while len(have) < count:
                have.append(cls(*key))
            return have[:count]


def _feed_video(clip: DeviceClip, rgb_src, slots, result) -> None:
    """Walks the video in order and uploads the frames the clip holds; result[0] = number of frames the video had (up to
    the clip's length), result[1] = an exception, if one ended the walk."""
    try:
        stream = torch.cuda.Stream(device=clip.rgb.device)
        got, k = 0, 0
        for f, frame in rgb_src.frames():
            if f >= clip.n_frames:
                break
            got = f + 1
            pos = clip._pos.get(f)
            if pos is None or frame is _io.SKIPPED:
                continue
            s = slots[k & 1]
            k += 1
            np.copyto(s.rgb_np, frame)
            with torch.cuda.stream(stream):
                clip.rgb[pos].copy_(s.rgb, non_blocking=True)
            stream.synchronize()
        result[0] = got
    except BaseException as e:  # handed to the caller's thread
        result[1] = e


def _fill(clip: DeviceClip, rgb_src, depth_src, mask_src, workers: Optional[int]) -> int:
    """Decodes and uploads every frame the clip holds; returns the number of frames the clip turned out to have (less
    than clip.n_frames only when an RGB video ends early)."""
    workers = workers or max(1, min(16, (os.cpu_count() or 2) - 1))
    m = len(clip.index)
    workers = min(workers, m)
    H, W = rgb_src.shape[:2]
    key = (H, W, None if clip.depth is None else clip.depth.dtype, clip.mask is not None)
    slots = _Staging.take(key, workers + 4)
    video, video_result = None, [clip.n_frames, None]
    if rgb_src.sequential:  # the last two slots' colour buffers belong to the thread that walks the video
        video = threading.Thread(target=_feed_video, args=(clip, rgb_src, slots[-2:], video_result), daemon=True)
        video.start()
    slots = slots[:-2]
    free = queue.SimpleQueue()
    for s in slots:
        free.put(s)

    def decode(k: int):
        f = clip.index[k]
        s = free.get()
        try:
            if not rgb_src.sequential:
                rgb_src.fill(f, s.rgb_np)
            if depth_src is not None:
                depth_src.fill(f, s.depth_np)
            if mask_src is not None:
                mask_src.fill(f, s.mask_np)
        except BaseException:
            free.put(s)
            raise
        return k, s

    copy_stream = torch.cuda.Stream(device=clip.rgb.device)
    with ThreadPoolExecutor(max_workers=workers) as pool:
        futs = [pool.submit(decode, k) for k in range(m)]
        try:
            for fut in as_completed(futs):
                k, s = fut.result()
                with torch.cuda.stream(copy_stream):
                    if not rgb_src.sequential:
                        clip.rgb[k].copy_(s.rgb, non_blocking=True)
                    if clip.depth is not None:
                        clip.depth[k].copy_(s.depth, non_blocking=True)
                    if clip.mask is not None:
                        clip.mask[k].copy_(s.mask, non_blocking=True)
                copy_stream.synchronize()  # 12 MB over the host link: a quarter of a millisecond, then the slot is free
                free.put(s)
        except BaseException:
            for fut in futs:
                fut.cancel()
            # unblock workers that wait for a slot so that the pool can shut down
            for _ in range(workers):
                free.put(slots[0])
            raise
    torch.cuda.current_stream(clip.rgb.device).wait_stream(copy_stream)
    if video is not None:
        video.join()  # its uploads are complete: every one of them was followed by a synchronisation of its stream
        if video_result[1] is not None:
            raise video_result[1]
    return video_result[0]
