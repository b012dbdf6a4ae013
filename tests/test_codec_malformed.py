"""Malformed depth (EXR) and mask (PNG) members: both decoders of egoprior.io -- the native one (csrc/egoprior_codec.c) and
the Python one -- must refuse them with OSError (the reference's reader turns that into an all-NaN frame,
vipe/utils/io.py:293-305) or decode them, never read outside the file.  The native decoder is also run over the same
files in a build with the address and undefined-behaviour sanitizers."""
import shutil
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "egox-egopriorrenderer_amd" / "csrc"


def _depth(H=37, W=29, seed=0):
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    z = (2.0 + 0.01 * u + 0.02 * v).astype(np.float16)
    z[20:, :5] = rng.random((H - 20, 5)).astype(np.float16)
    return z


def _table_pos(buf: bytes) -> int:
    """Offset of the chunk table of an EXR file written by egoprior.io.encode_exr_half_z (end of the header)."""
    pos = 8
    while buf[pos] != 0:
        e = buf.index(b"\0", pos)
        e2 = buf.index(b"\0", e + 1)
        size, = struct.unpack_from("<i", buf, e2 + 1)
        pos = e2 + 5 + size
    return pos + 1


def exr_mutations():
    from egoprior import io as eio
    z = _depth()
    out = []
    for comp in (0, 1, 2, 3):
        good = eio.encode_exr_half_z(z, compression=comp)
        n = len(good)
        tp = _table_pos(good)
        first, = struct.unpack_from("<Q", good, tp)
        out.append((f"c{comp}_good", good))
        for cut in sorted({0, 1, 7, 8, 15, 16, tp - 1, tp, tp + 4, tp + 8, first, first + 4, first + 8, first + 9,
                           n // 2, n - 9, n - 8, n - 2, n - 1}):
            if 0 <= cut < n:
                out.append((f"c{comp}_cut{cut}", good[:cut]))
        # tampered offset table: offsets near 2^64 (additions on them wrap), at and around the end of the file, into the header
        for k, off in enumerate((2 ** 64 - 1, 2 ** 64 - 4, 2 ** 64 - 8, 2 ** 64 - 9, 2 ** 63, n, n - 1, n - 4, n - 7, n - 8,
                                 n - 9, n + 1000, 0, 8, tp)):
            b = bytearray(good)
            struct.pack_into("<Q", b, tp, off)
            out.append((f"c{comp}_off{k}", bytes(b)))
            b = bytearray(good)
            struct.pack_into("<Q", b, tp + 8, off)  # a later chunk
            out.append((f"c{comp}_off2_{k}", bytes(b)))
        # tampered chunk header: row number and payload size
        for k, (y, size) in enumerate(((None, -1), (None, -2 ** 31), (None, 2 ** 31 - 1), (None, 0), (None, 1), (None, 3),
                                       (-1, None), (10 ** 6, None), (-2 ** 31, None), (36, None), (37, None))):
            b = bytearray(good)
            if y is not None:
                struct.pack_into("<i", b, first, y)
            if size is not None:
                struct.pack_into("<i", b, first + 4, size)
            out.append((f"c{comp}_hdr{k}", bytes(b)))
        # the last chunk's size claims more than the file holds / the file ends inside the last chunk's payload
        last, = struct.unpack_from("<Q", good, tp + 8 * ((37 + (16 if comp == 3 else 1) - 1) // (16 if comp == 3 else 1) - 1))
        for trim in (1, 2, 5):
            out.append((f"c{comp}_lasttrim{trim}", good[:n - trim]))
        b = bytearray(good)
        sz, = struct.unpack_from("<i", good, last + 4)
        struct.pack_into("<i", b, last + 4, sz + 1)
        out.append((f"c{comp}_lastplus1", bytes(b)))
        # damaged payload bytes (compressed streams: the inflate / RLE decoder sees garbage)
        rng = np.random.default_rng(comp)
        for k in range(6):
            b = bytearray(good)
            for p in rng.integers(first + 8, n, size=4):
                b[int(p)] ^= int(rng.integers(1, 256))
            out.append((f"c{comp}_noise{k}", bytes(b)))
    return out


def png_mutations():
    from egoprior import io as eio
    rng = np.random.default_rng(5)
    img = (rng.random((33, 41)) < 0.3).astype(np.uint8) * rng.integers(1, 5, (33, 41)).astype(np.uint8)
    good = eio.encode_png_gray(img)
    n = len(good)
    out = [("png_good", good)]
    for cut in (0, 7, 8, 20, 32, 33, 40, n // 2, n - 13, n - 12, n - 4, n - 1):
        out.append((f"png_cut{cut}", good[:cut]))
    idat = good.index(b"IDAT")
    for k, ln in enumerate((2 ** 32 - 1, 2 ** 31, n, n - idat, 0, 1)):
        b = bytearray(good)
        struct.pack_into(">I", b, idat - 4, ln)
        out.append((f"png_len{k}", bytes(b)))
    for k in range(6):
        b = bytearray(good)
        for p in rng.integers(idat + 4, n - 12, size=3):
            b[int(p)] ^= int(rng.integers(1, 256))
        out.append((f"png_noise{k}", bytes(b)))
    return out


def _outcome(fn, buf, **kw):
    try:
        return fn(buf, **kw)
    except OSError:
        return None


def test_malformed_exr_members_are_refused_by_both_decoders():
    from egoprior import io as eio
    z = _depth()
    refused = 0
    for name, buf in exr_mutations():
        a = _outcome(eio.decode_exr_half_z, buf, native=True)
        b = _outcome(eio.decode_exr_half_z, buf, native=False)
        assert (a is None) == (b is None), f"{name}: the native and the Python decoder disagree on whether the file reads"
        if a is None:
            refused += 1
            continue
        assert a.shape == z.shape and a.dtype == np.float16, name
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"{name}: decoders differ"
        if name.endswith("_good"):
            assert np.array_equal(a.view(np.uint16), z.view(np.uint16))
    assert refused > 100
    # the two cases of the report: an offset whose `+ 8` wraps, and an uncompressed chunk shorter than its rows
    good = eio.encode_exr_half_z(z, compression=0)
    tp = _table_pos(good)
    first, = struct.unpack_from("<Q", good, tp)
    b = bytearray(good)
    struct.pack_into("<Q", b, tp, 2 ** 64 - 4)
    for native in (True, False):
        with pytest.raises(OSError):
            eio.decode_exr_half_z(bytes(b), native=native)
    b = bytearray(good)
    struct.pack_into("<i", b, first + 4, 3)
    for native in (True, False):
        with pytest.raises(OSError):
            eio.decode_exr_half_z(bytes(b), native=native)


def test_malformed_png_members_are_refused_by_both_decoders():
    from egoprior import io as eio
    for name, buf in png_mutations():
        a = _outcome(eio.decode_png_gray, buf, native=True)
        b = _outcome(eio.decode_png_gray, buf, native=False)
        # (a damaged stream whose checksum zlib still accepts cannot happen with these few flipped bytes)
        assert (a is None) == (b is None), f"{name}: the native and the Python decoder disagree on whether the file reads"
        if a is not None:
            assert np.array_equal(a, b), name


def test_a_damaged_member_becomes_an_all_nan_frame(tmp_path):
    """vipe/utils/io.py:293-305 through the zip reader, with the tampered offset table."""
    import zipfile
    from egoprior import io as eio
    z = _depth()
    good = eio.encode_exr_half_z(z)
    tp = _table_pos(good)
    bad = bytearray(good)
    struct.pack_into("<Q", bad, tp, 2 ** 64 - 4)
    p = tmp_path / "depth.zip"
    with zipfile.ZipFile(p, "w") as zf:
        zf.writestr("00000.exr", good)
        zf.writestr("00001.exr", bytes(bad))
    frames = [d for _, d in eio.read_depth_artifacts(p)]
    assert len(frames) == 2
    assert np.array_equal(np.asarray(frames[0], dtype=np.float16).view(np.uint16), z.view(np.uint16))
    assert np.isnan(np.asarray(frames[1])).all()


def test_native_decoders_under_the_sanitizers(tmp_path):
    """The same files through a build of the codec with -fsanitize=address,undefined: exact-size heap buffers, so any
    access outside the file or the output aborts the driver."""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "codec_asan"
    cmd = [gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
           str(CSRC / "egoprior_codec.c"), str(ROOT / "tests" / "native" / "codec_asan_driver.c"), "-lz", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available here: " + r.stderr[-300:])
    files = []
    for name, buf in exr_mutations():
        f = tmp_path / f"{name}.exr"
        f.write_bytes(buf)
        files.append(str(f))
    for name, buf in png_mutations():
        f = tmp_path / f"{name}.png"
        f.write_bytes(buf)
        files.append(str(f))
    r = subprocess.run([str(exe)] + files, capture_output=True, text=True,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files)
    ok = [ln for ln in lines if ln.endswith(" 0 0")]
    assert any("_good" in ln for ln in ok) and len(ok) < len(lines)
