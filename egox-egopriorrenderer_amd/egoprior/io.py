"""ViPE artifact layout and readers for the ego-prior renderer.

Mirrors the parts of the reference's vipe/utils/io.py that the render path touches (ArtifactPath :41-146,
read_pose_artifacts :169-171, read_intrinsics_artifacts :221-234, read_rgb_artifacts :244-251, read_depth_artifacts
:282-312, read_instance_artifacts :315-327) with two differences that matter on a machine without imageio / OpenEXR /
OpenCV:

  * every artifact also has an array twin that needs no codec: rgb/<name>.npy (F,H,W,3) uint8, depth/<name>.npy (F,H,W)
    float16 or float32 (or the per-frame files depth/<name>/depth_%06d.npy of scripts/convert_depth_zip_to_npy.py),
    mask/<name>.npy (F,H,W) uint8.  The twin is used when the codec file is absent; synthetic
    sequences are written this way (write_synthetic_artifacts).
  * instance-mask PNGs are decoded by a small pure-Python decoder (8/16-bit greyscale, non-interlaced) when OpenCV is
    missing, and depth EXRs by a built-in reader of the subset ViPE writes (single-part scan-line file, HALF channel Z,
    compression none / RLE / ZIPS / ZIP) when OpenEXR is missing; RGB mp4s need imageio -- if it is missing and there is
    no twin, the reader raises with a message that says so.

Poses are returned as the (F,4,4) matrices stored in the npz (the reference round-trips them through a normalised
quaternion, vipe/utils/geometry.py:339-344, which changes a valid pose by < 1e-6).
"""
from __future__ import annotations

import ctypes as _C
import io as _io
import struct
import threading
import zipfile
import zlib
from dataclasses import dataclass
from pathlib import Path
from typing import Iterator, Optional

import numpy as np


@dataclass
class ArtifactPath:
    base_path: Path
    artifact_name: str

    @property
    def rgb_path(self) -> Path:
        return self.base_path / "rgb" / f"{self.artifact_name}.mp4"

    @property
    def pose_path(self) -> Path:
        return self.base_path / "pose" / f"{self.artifact_name}.npz"

    @property
    def depth_path(self) -> Path:
        return self.base_path / "depth" / f"{self.artifact_name}.zip"

    @property
    def intrinsics_path(self) -> Path:
        return self.base_path / "intrinsics" / f"{self.artifact_name}.npz"

    @property
    def camera_type_path(self) -> Path:
        return self.base_path / "intrinsics" / f"{self.artifact_name}_camera.txt"

    @property
    def mask_path(self) -> Path:
        return self.base_path / "mask" / f"{self.artifact_name}.zip"

    @classmethod
    def glob_artifacts(cls, base_path: Path, use_video: bool = False) -> Iterator["ArtifactPath"]:
        base_path = Path(base_path)
        if use_video:
            seen = set()
            for pat in ("*.mp4", "*.npy"):
                for p in sorted((base_path / "rgb").glob(pat)):
                    if p.stem not in seen:
                        seen.add(p.stem)
                        yield cls(base_path, p.stem)
        else:
            for p in sorted((base_path / "vipe").glob("*_info.pkl")):
                yield cls(base_path, p.stem.replace("_info", ""))


def _twin(path: Path) -> Path:
    return path.with_suffix(".npy")


# ----------------------------------------------------------------------------------------------------------------------
# Native decoders (csrc/egoprior_codec.c): the same two formats as decode_exr_half_z / decode_png_gray below, decoded
# straight into a caller-owned buffer with the GIL released.  Optional: without the library the Python decoders run.
# ----------------------------------------------------------------------------------------------------------------------
_CODEC_PATH = Path(__file__).resolve().parent.parent / "csrc" / "libegoprior_codec.so"
_codec_lib = None
_codec_lock = threading.Lock()


def native_codec():
    """The ctypes handle of libegoprior_codec.so, or None when it has not been built."""
    global _codec_lib
    if _codec_lib is None:
        with _codec_lock:
            if _codec_lib is None:
                if not _CODEC_PATH.exists():
                    _codec_lib = False
                else:
                    L = _C.CDLL(str(_CODEC_PATH))
                    i32p = _C.POINTER(_C.c_int32)
                    for name in ("epc_exr_half_z_size", "epc_png_gray8_size"):
                        getattr(L, name).argtypes = [_C.c_char_p, _C.c_size_t, i32p, i32p]
                        getattr(L, name).restype = _C.c_int
                    for name in ("epc_exr_half_z_decode", "epc_png_gray8_decode"):
                        getattr(L, name).argtypes = [_C.c_char_p, _C.c_size_t, _C.c_void_p, _C.c_int32, _C.c_int32]
                        getattr(L, name).restype = _C.c_int
                    _codec_lib = L
    return _codec_lib or None


def _native_decode(kind: str, buf: bytes, dtype, out: Optional[np.ndarray]):
    """The decoded (H, W) array (`out` itself when given), or None when the library is missing or declines the file.
    A given `out` of the wrong shape / dtype / layout is an error of the caller."""
    L = native_codec()
    if L is None:
        return None
    h, w = _C.c_int32(), _C.c_int32()
    if getattr(L, f"epc_{kind}_size")(buf, len(buf), _C.byref(h), _C.byref(w)) != 0:
        return None
    if out is None:
        out = np.empty((h.value, w.value), dtype=dtype)
    elif out.shape != (h.value, w.value) or out.dtype != np.dtype(dtype) or not out.flags.c_contiguous:
        raise ValueError(f"out must be a C-contiguous ({h.value},{w.value}) {np.dtype(dtype).name} array")
    if getattr(L, f"epc_{kind}_decode")(buf, len(buf), out.ctypes.data, h.value, w.value) != 0:
        return None
    return out


# ----------------------------------------------------------------------------------------------------------------------
class _Skipped:
    """Stands for a frame the caller did not ask for (see `wanted` of the readers)."""


SKIPPED = _Skipped()


def _frame_index(name: str) -> int:
    return int(name.split(".")[0])


def _decode_ahead(z: zipfile.ZipFile, names, decode, workers: Optional[int] = None, wanted=None):
    """Yields (name, decode(bytes of the member)) in the order of `names`; members whose frame index is not in `wanted`
    (when given) are neither read nor decoded and yield SKIPPED.  The members are read from the archive one
    after the other (the archive object is not thread-safe) and decoded by a few worker threads, a bounded number of
    frames ahead of the consumer: inflate and the large numpy operations of the built-in decoders release the GIL, so
    a 1080p clip decodes several times faster than frame by frame.  An exception raised by `decode` is delivered in
    place of the result, so that the caller can treat that frame the way the reference does."""
    import os
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    def guarded(data):
        try:
            return decode(data)
        except Exception as e:  # handed to the consumer, in order
            return e

    workers = workers or max(1, min(8, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        ahead = deque()
        for name in names:
            if wanted is not None and _frame_index(name) not in wanted:
                ahead.append((name, None))
            else:
                ahead.append((name, pool.submit(guarded, z.read(name))))
            if len(ahead) > 2 * workers:
                n, fut = ahead.popleft()
                yield n, (SKIPPED if fut is None else fut.result())
        while ahead:
            n, fut = ahead.popleft()
            yield n, (SKIPPED if fut is None else fut.result())


def read_pose_artifacts(npz_file_path: Path):
    """(inds (F,), c2w (F,4,4) float32) -- OpenCV camera-to-world matrices."""
    data = np.load(npz_file_path)
    return data["inds"], np.asarray(data["data"], dtype=np.float32)


def read_intrinsics_artifacts(intr_file_path: Path, camera_file_path: Optional[Path] = None):
    """(inds, intrinsics (F,4) float32 [fx,fy,cx,cy], camera type names).  A missing camera-type file means PINHOLE."""
    data = np.load(intr_file_path)
    inds, intr = data["inds"], np.asarray(data["data"], dtype=np.float32)
    if camera_file_path is None or not Path(camera_file_path).exists():
        assert intr.shape[1] == 4
        types = ["PINHOLE"] * intr.shape[0]
    else:
        with open(camera_file_path) as f:
            types = [line.split(":")[1].strip() for line in f.readlines() if ":" in line]
    return inds, intr, types


def read_rgb_artifacts(rgb_file_path: Path, wanted=None) -> Iterator[tuple[int, np.ndarray]]:
    """Yields (frame index, (H,W,3) uint8).  (The reference yields uint8/255 as float32; the renderer only ever needs
    the bytes back: (rgb*255).astype(uint8) is the identity on them.)  wanted: a container of frame indices; the other
    frames yield SKIPPED (a .npy twin is then not even read for them; a video still has to be decoded in order)."""
    rgb_file_path = Path(rgb_file_path)
    if rgb_file_path.exists() and rgb_file_path.suffix != ".npy":
        try:
            import imageio  # noqa: F401
        except ImportError as e:
            if not _twin(rgb_file_path).exists():
                raise ImportError(f"reading {rgb_file_path} needs imageio (ffmpeg); no {_twin(rgb_file_path).name} "
                                  "twin found") from e
        else:
            reader = imageio.get_reader(rgb_file_path, "ffmpeg")
            for i, frame in enumerate(reader):
                yield i, (SKIPPED if (wanted is not None and i not in wanted) else np.asarray(frame, dtype=np.uint8))
            return
    tw = rgb_file_path if rgb_file_path.suffix == ".npy" else _twin(rgb_file_path)
    if not tw.exists():
        raise FileNotFoundError(str(rgb_file_path))
    arr = np.load(tw, mmap_mode="r")
    for i in range(arr.shape[0]):
        yield i, (SKIPPED if (wanted is not None and i not in wanted) else np.asarray(arr[i], dtype=np.uint8))


def read_depth_artifacts(zip_file_path: Path, keep_half: bool = False, wanted=None) -> Iterator[tuple[int, np.ndarray]]:
    """Yields (frame index, (H,W) float32 metric depth).  EXR channel Z is HALF; a frame that fails to load yields NaN.
    keep_half: hand float16 data on as float16 (half the upload; the GPU kernels widen it, which is exact) instead of
    widening it here as vipe/utils/io.py:277 does.  wanted: a container of frame indices; the other frames are not
    decoded and yield SKIPPED in place of the array (a rank of a multi-GPU run only needs its own block)."""
    zip_file_path = Path(zip_file_path)
    if zip_file_path.exists():
        try:
            import OpenEXR
        except ImportError:
            OpenEXR = None  # the built-in reader covers the files ViPE writes (one HALF channel Z, scan lines)
        vw = vh = 0
        with zipfile.ZipFile(zip_file_path, "r") as z:
            names = sorted(z.namelist())
            if OpenEXR is None:
                for name, half in _decode_ahead(z, names, decode_exr_half_z, wanted=wanted):
                    idx = int(name.split(".")[0])
                    if half is SKIPPED:
                        yield idx, SKIPPED
                        continue
                    if isinstance(half, OSError):
                        assert vw > 0 and vh > 0
                        yield idx, np.full((vh, vw), np.nan, dtype=np.float32)
                        continue
                    if isinstance(half, Exception):
                        raise half
                    vh, vw = half.shape
                    yield idx, (half if keep_half else half.astype(np.float32))
                return
            for name in names:
                idx = int(name.split(".")[0])
                if wanted is not None and idx not in wanted:
                    yield idx, SKIPPED
                    continue
                with z.open(name) as f:
                    try:
                        exr = OpenEXR.InputFile(f)
                    except OSError:
                        assert vw > 0 and vh > 0
                        yield idx, np.full((vh, vw), np.nan, dtype=np.float32)
                        continue
                    dw = exr.header()["dataWindow"]
                    vw, vh = dw.max.x - dw.min.x + 1, dw.max.y - dw.min.y + 1
                    half = np.frombuffer(exr.channels(["Z"])[0], dtype=np.float16).reshape(vh, vw)
                    yield idx, (half.copy() if keep_half else half.astype(np.float32))
        return
    tw = _twin(zip_file_path)
    if not tw.exists():
        # per-frame arrays as scripts/convert_depth_zip_to_npy.py:108-164 writes them: <zip stem>/depth_%06d.npy, float32
        frame_dir = zip_file_path.with_suffix("")
        files = sorted(frame_dir.glob("depth_*.npy")) if frame_dir.is_dir() else []
        if not files:
            raise FileNotFoundError(str(zip_file_path))
        for f in files:
            i = int(f.stem.split("_")[-1])
            if wanted is not None and i not in wanted:
                yield i, SKIPPED
                continue
            a = np.load(f)
            yield i, (a.copy() if (keep_half and a.dtype == np.float16) else a.astype(np.float32))
        return
    arr = np.load(tw, mmap_mode="r")
    for i in range(arr.shape[0]):
        if wanted is not None and i not in wanted:
            yield i, SKIPPED
            continue
        a = np.asarray(arr[i])
        yield i, (a.copy() if (keep_half and a.dtype == np.float16) else a.astype(np.float32))


# ----------------------------------------------------------------------------------------------------------------------
# OpenEXR, the subset ViPE writes: one scan-line part, one HALF channel named Z (vipe/utils/io.py:270-279)
# ----------------------------------------------------------------------------------------------------------------------
_EXR_MAGIC = 20000630
_EXR_NO, _EXR_RLE, _EXR_ZIPS, _EXR_ZIP = 0, 1, 2, 3
_EXR_LINES = {_EXR_NO: 1, _EXR_RLE: 1, _EXR_ZIPS: 1, _EXR_ZIP: 16}


def _exr_unfilter(t: np.ndarray) -> np.ndarray:
    """Undo the byte predictor and the even / odd byte split OpenEXR applies before deflate (ImfZip: every byte is
    stored as its difference to the previous one plus 128; the first half of the buffer holds the bytes of even
    position, the second half those of odd position)."""
    n = t.size
    d = np.cumsum(t, dtype=np.uint8)  # wraps modulo 256, which is the arithmetic of the format
    d[1::2] ^= 0x80                   # minus 128 * position: 128 at odd positions, 0 at even ones (mod 256)
    out = np.empty(n, dtype=np.uint8)
    half = (n + 1) // 2
    out[0::2] = d[:half]
    out[1::2] = d[half:]
    return out


def _exr_filter(raw: np.ndarray) -> np.ndarray:
    n = raw.size
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 255).astype(np.uint8)


def _exr_unrle(src: bytes, n_out: int) -> np.ndarray:
    """OpenEXR's run-length code: a count byte c >= 0 repeats the next byte c + 1 times, c < 0 copies -c bytes."""
    out = bytearray()
    i = 0
    while i < len(src) and len(out) < n_out:
        c = src[i] - 256 if src[i] > 127 else src[i]
        i += 1
        if c < 0:
            out += src[i:i - c]
            i += -c
        else:
            out += bytes([src[i]]) * (c + 1)
            i += 1
    return np.frombuffer(bytes(out), dtype=np.uint8)


def exr_size(buf: bytes) -> tuple[int, int]:
    """(H, W) of the data window of an OpenEXR file; OSError like decode_exr_half_z."""
    L = native_codec()
    if L is not None:
        h, w = _C.c_int32(), _C.c_int32()
        if L.epc_exr_half_z_size(buf, len(buf), _C.byref(h), _C.byref(w)) == 0:
            return h.value, w.value
    return decode_exr_half_z(buf, native=False).shape


def _exr_rle(src: bytes) -> bytes:
    """The inverse of _exr_unrle (runs of 3..128 equal bytes as a run, everything else as literal stretches <= 127)."""
    out, i, n = bytearray(), 0, len(src)
    lit = bytearray()

    def flush():
        for k in range(0, len(lit), 127):
            part = lit[k:k + 127]
            out.append(256 - len(part))
            out.extend(part)
        lit.clear()

    while i < n:
        j = i
        while j < n and j - i < 128 and src[j] == src[i]:
            j += 1
        if j - i >= 3:
            flush()
            out.append(j - i - 1)
            out.append(src[i])
        else:
            lit.extend(src[i:j])
        i = j
    flush()
    return bytes(out)


def decode_exr_half_z(buf: bytes, out: Optional[np.ndarray] = None, native: bool = True) -> np.ndarray:
    """(H, W) float16 of channel Z of a single-part scan-line OpenEXR file with compression none, RLE, ZIPS or ZIP.
    Raises OSError for anything that is not such a file (what OpenEXR.InputFile raises, and what the reference's reader
    catches to substitute an all-NaN frame, vipe/utils/io.py:293-305).  out: a C-contiguous (H, W) float16 array to
    decode into (returned).  native=False forces the Python decoder (the two agree bit for bit; the tests compare them)."""
    if native:
        got = _native_decode("exr_half_z", buf, np.float16, out)
        if got is not None:
            return got
    res = _decode_exr_half_z_py(buf)  # also the path that words the error for a file the native decoder declined
    if out is not None:
        if out.shape != res.shape or out.dtype != np.float16:
            raise ValueError(f"out must be a {res.shape} float16 array")
        out[...] = res
        return out
    return res


def _decode_exr_half_z_py(buf: bytes) -> np.ndarray:
    try:
        magic, version = struct.unpack_from("<iI", buf, 0)
        if magic != _EXR_MAGIC or (version & 0xFF) != 2 or (version & 0x1A00):  # tiled / deep / multi-part bits
            raise ValueError("not a single-part scan-line OpenEXR file")
        pos, attrs = 8, {}
        while buf[pos] != 0:
            e = buf.index(b"\0", pos)
            name = buf[pos:e].decode("latin1")
            e2 = buf.index(b"\0", e + 1)
            size, = struct.unpack_from("<i", buf, e2 + 1)
            attrs[name] = buf[e2 + 5:e2 + 5 + size]
            pos = e2 + 5 + size
        pos += 1
        xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"])
        W, H = xmax - xmin + 1, ymax - ymin + 1
        comp = attrs["compression"][0]
        if comp not in _EXR_LINES:
            raise ValueError(f"unsupported EXR compression {comp}")
        channels, c = [], attrs["channels"]
        q = 0
        while c[q] != 0:
            e = c.index(b"\0", q)
            ptype, _, xs, ys = struct.unpack_from("<iB3xii", c, e + 1)
            channels.append((c[q:e].decode("latin1"), ptype, xs, ys))
            q = e + 17
        if any(xs != 1 or ys != 1 for _, _, xs, ys in channels):
            raise ValueError("subsampled channels are not supported")
        names = [n for n, _, _, _ in channels]  # stored in alphabetical order
        if "Z" not in names or channels[names.index("Z")][1] != 1:
            raise ValueError("no HALF channel named Z")
        bpp = [4 if t != 1 else 2 for _, t, _, _ in channels]
        z_off = sum(bpp[:names.index("Z")]) * W
        line_bytes = sum(bpp) * W
        lines = _EXR_LINES[comp]
        n_chunks = (H + lines - 1) // lines
        offsets = struct.unpack_from(f"<{n_chunks}Q", buf, pos)
        out = np.empty((H, W), dtype=np.float16)
        for c_no, off in enumerate(offsets):
            if off > len(buf) - 8:
                raise ValueError("chunk offset beyond the end of the file")
            y, size = struct.unpack_from("<ii", buf, off)
            if y != ymin + c_no * lines or size < 0:  # the table lists the chunks by increasing row
                raise ValueError("corrupt chunk header")
            rows = min(lines, ymax - y + 1)
            raw_size = rows * line_bytes
            data = buf[off + 8:off + 8 + size]
            if size > raw_size or len(data) != size:
                raise ValueError("corrupt chunk")
            if size == raw_size:  # uncompressed file, or a chunk that did not shrink and is stored as it is
                raw = np.frombuffer(data, dtype=np.uint8)
            elif comp == _EXR_NO:
                raise ValueError("uncompressed chunk shorter than its rows")
            elif comp == _EXR_RLE:
                raw = _exr_unfilter(_exr_unrle(data, raw_size))
            else:
                raw = _exr_unfilter(np.frombuffer(zlib.decompress(data), dtype=np.uint8))
            if raw.size != raw_size:
                raise ValueError("chunk has the wrong size")
            raw = raw.reshape(rows, line_bytes)
            out[y - ymin:y - ymin + rows] = raw[:, z_off:z_off + 2 * W].copy().view("<f2")
        return out
    except (ValueError, KeyError, IndexError, struct.error, zlib.error) as e:
        raise OSError(f"cannot read EXR data: {e}") from e


def encode_exr_half_z(depth: np.ndarray, compression: int = _EXR_ZIP) -> bytes:
    """The file OpenEXR.OutputFile writes for Header(width, height) with channels {"Z": HALF} (the reference's writer,
    vipe/utils/io.py:270-279; the library's default compression is ZIP, 16 scan lines per chunk)."""
    half = np.ascontiguousarray(depth, dtype="<f2")
    H, W = half.shape

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value

    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    head = struct.pack("<iI", _EXR_MAGIC, 2)
    head += attr("channels", "chlist", b"Z\0" + struct.pack("<iB3xii", 1, 0, 1, 1) + b"\0")
    head += attr("compression", "compression", bytes([compression]))
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    head += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = _EXR_LINES[compression]
    chunks = []
    for y in range(0, H, lines):
        raw = half[y:y + lines].view(np.uint8).reshape(-1)
        data = raw.tobytes()
        if compression in (_EXR_ZIP, _EXR_ZIPS):
            packed = zlib.compress(_exr_filter(raw).tobytes())
            if len(packed) < len(data):
                data = packed
        elif compression == _EXR_RLE:
            packed = _exr_rle(_exr_filter(raw).tobytes())
            if len(packed) < len(data):
                data = packed
        elif compression != _EXR_NO:
            raise ValueError("the writer knows compression none, RLE, ZIPS and ZIP")
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    table_at = len(head)
    pos = table_at + 8 * len(chunks)
    table = b""
    for ch in chunks:
        table += struct.pack("<Q", pos)
        pos += len(ch)
    return head + table + b"".join(chunks)


def decode_png_gray(buf: bytes, out: Optional[np.ndarray] = None, native: bool = True) -> np.ndarray:
    """Minimal PNG decoder: 8- or 16-bit greyscale (colour type 0), non-interlaced.  8-bit files go through the native
    decoder when it is built (native=False forces the Python one; they agree bit for bit); out: a C-contiguous (H, W)
    uint8 array to decode an 8-bit file into (returned)."""
    if native:
        got = _native_decode("png_gray8", buf, np.uint8, out)
        if got is not None:
            return got
    res = _decode_png_gray_py(buf)
    if out is not None:
        if out.shape != res.shape or out.dtype != res.dtype:
            raise ValueError(f"out must be a {res.shape} {res.dtype.name} array")
        out[...] = res
        return out
    return res


def _decode_png_gray_py(buf: bytes) -> np.ndarray:
    """OSError for anything that is not a readable greyscale PNG (truncated, damaged stream, other colour types)."""
    try:
        return _decode_png_gray_checked(buf)
    except (ValueError, TypeError, IndexError, struct.error, zlib.error) as e:
        raise OSError(f"cannot read PNG data: {e}") from e


def _decode_png_gray_checked(buf: bytes) -> np.ndarray:
    if buf[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, hdr = 8, [], None
    while pos + 12 <= len(buf):  # (a file that ends after its last whole chunk is read as far as it goes)
        n, typ = struct.unpack(">I4s", buf[pos:pos + 8])
        if pos + 12 + n > len(buf):
            raise ValueError("chunk runs past the end of the file")
        body = buf[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None or not idat:
        raise ValueError("no IHDR / IDAT chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    if ctype != 0 or depth not in (8, 16) or interlace != 0:
        raise ValueError(f"unsupported PNG (colour type {ctype}, depth {depth}, interlace {interlace}); install opencv")
    bpp = depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, stride + 1)
    ftypes = raw[:, 0]
    if not ftypes.any():  # every row stored as it is: nothing to undo
        out = np.ascontiguousarray(raw[:, 1:])
    else:
        out = np.zeros((h, stride), dtype=np.uint8)
        prev = np.zeros(stride, dtype=np.uint8)
        for y in range(h):
            ft = int(ftypes[y])
            line = raw[y, 1:]
            if ft == 0:
                cur = line
            elif ft == 2:  # Up: uint8 arithmetic wraps modulo 256, which is the arithmetic of the format
                cur = line + prev
            elif ft == 1:  # Sub: a running sum along the row, one per byte lane of a pixel
                cur = np.empty(stride, dtype=np.uint8)
                for c in range(bpp):
                    cur[c::bpp] = np.cumsum(line[c::bpp], dtype=np.uint8)
            elif ft in (3, 4):  # Average / Paeth depend on the pixel just decoded: one step per byte
                cur = np.zeros(stride, dtype=np.uint8)
                ln, pv = line.tolist(), prev.tolist()
                res = [0] * stride
                for x in range(stride):
                    left = res[x - bpp] if x >= bpp else 0
                    up = pv[x]
                    if ft == 3:
                        pred = (left + up) >> 1
                    else:
                        ul = pv[x - bpp] if x >= bpp else 0
                        pa, pb, pc = abs(up - ul), abs(left - ul), abs(left + up - 2 * ul)
                        pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
                    res[x] = (ln[x] + pred) & 255
                cur[:] = res
            else:
                raise ValueError("bad PNG filter")
            out[y] = cur
            prev = out[y]
    if depth == 16:
        o = out.reshape(h, w, 2).astype(np.uint16)
        return o[..., 0] * 256 + o[..., 1]
    return out


def decode_instance_png(data: bytes) -> np.ndarray:
    """One member of the mask archive as (H, W) uint8: OpenCV's decoder when it is importable (what the reference calls,
    vipe/utils/io.py:325), otherwise the built-in greyscale decoders."""
    try:
        import cv2
    except ImportError:
        return decode_png_gray(data).astype(np.uint8)
    return np.asarray(cv2.imdecode(np.frombuffer(data, dtype=np.uint8), cv2.IMREAD_UNCHANGED)).astype(np.uint8)


def read_instance_artifacts(zip_file_path: Path, wanted=None) -> Iterator[tuple[int, np.ndarray]]:
    """Yields (frame index, (H,W) uint8 instance ids), 0 = background."""
    zip_file_path = Path(zip_file_path)
    if zip_file_path.exists():
        with zipfile.ZipFile(zip_file_path, "r") as z:
            for name, m in _decode_ahead(z, sorted(z.namelist()), decode_instance_png, wanted=wanted):
                if m is SKIPPED:
                    yield int(name.split(".")[0]), SKIPPED
                    continue
                if isinstance(m, Exception):
                    raise m
                yield int(name.split(".")[0]), m
        return
    tw = _twin(zip_file_path)
    if not tw.exists():
        raise FileNotFoundError(str(zip_file_path))
    arr = np.load(tw, mmap_mode="r")
    for i in range(arr.shape[0]):
        yield i, (SKIPPED if (wanted is not None and i not in wanted) else np.asarray(arr[i], dtype=np.uint8))


# ----------------------------------------------------------------------------------------------------------------------
def encode_png_gray(img: np.ndarray) -> bytes:
    """8-bit greyscale PNG (filter 0), used to write synthetic instance masks in the reference's container format."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    raw = np.concatenate([np.zeros((h, 1), np.uint8), img], axis=1).tobytes()
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def encode_png_rgb(img: np.ndarray) -> bytes:
    """8-bit RGB PNG (filter 0) of an (H, W, 3) uint8 image."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 3)], axis=1).tobytes()
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_synthetic_artifacts(base: Path, name: str, depth, rgb, mask, intr, c2w, mask_as_png_zip: bool = True,
                              camera_type: str = "PINHOLE", depth_as_exr_zip: bool = False):
    """Writes a sequence in the ViPE layout using the array twins for rgb and depth (no codec needed) and, by default,
    the reference's zip-of-PNG container for the instance masks.  depth is stored as float16 like the EXR HALF files."""
    base = Path(base)
    depth, rgb, mask = np.asarray(depth), np.asarray(rgb), None if mask is None else np.asarray(mask)
    Fn = depth.shape[0]
    for sub in ("rgb", "depth", "pose", "intrinsics", "mask"):
        (base / sub).mkdir(parents=True, exist_ok=True)
    np.save(base / "rgb" / f"{name}.npy", rgb.astype(np.uint8))
    if depth_as_exr_zip:  # the reference's container: a zip of %05d.exr files, channel Z HALF (vipe/utils/io.py:268-279)
        with zipfile.ZipFile(base / "depth" / f"{name}.zip", "w", zipfile.ZIP_DEFLATED) as z:
            for i in range(Fn):
                z.writestr(f"{i:05d}.exr", encode_exr_half_z(depth[i].astype(np.float16)))
    else:
        np.save(base / "depth" / f"{name}.npy", depth.astype(np.float16))
    c2w = np.asarray(c2w, dtype=np.float32)
    if c2w.ndim == 2:
        c2w = np.repeat(c2w[None], Fn, axis=0)
    np.savez(base / "pose" / f"{name}.npz", data=c2w, inds=np.arange(Fn))
    intr = np.asarray(intr, dtype=np.float32)
    if intr.ndim == 1:
        intr = np.repeat(intr[None], Fn, axis=0)
    np.savez(base / "intrinsics" / f"{name}.npz", data=intr, inds=np.arange(Fn))
    with open(base / "intrinsics" / f"{name}_camera.txt", "w") as f:
        for i in range(Fn):
            f.write(f"{i}: {camera_type}\n")
    if mask is not None:
        if mask_as_png_zip:
            with zipfile.ZipFile(base / "mask" / f"{name}.zip", "w", zipfile.ZIP_DEFLATED) as z:
                for i in range(Fn):
                    z.writestr(f"{i:05d}.png", encode_png_gray(mask[i]))
        else:
            np.save(base / "mask" / f"{name}.npy", mask.astype(np.uint8))
    return ArtifactPath(base, name)


def write_video(path: Path, frames, fps: int = 30) -> Path:
    """ego_Prior.mp4 exactly as the reference writes it (imageio, libx264, quality 8, yuv420p; scripts/
    render_vipe_pointcloud.py:1444) when imageio is importable; otherwise the frames go to <stem>.npy next to the
    requested path and that path is returned."""
    path = Path(path)
    try:
        import imageio
    except ImportError:
        alt = path.with_suffix(".npy")
        np.save(alt, np.stack([np.asarray(f, dtype=np.uint8) for f in frames], axis=0))
        return alt
    with imageio.get_writer(str(path), fps=fps, codec="libx264", quality=8, pixelformat="yuv420p") as w:
        for f in frames:
            w.append_data(np.asarray(f, dtype=np.uint8))
    return path
