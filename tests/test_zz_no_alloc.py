"""The entry point that allocates nothing (egoprior_unproject_frame_noalloc: static device scratch instead of a
caller-owned workspace) against egoprior_unproject_frame, through the C ABI: same compacted points bit for bit, on two
calls in a row, the second with a smaller frame (what the first call left in the scratch must not matter)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _both(engine, H, W, s, select, frame_id, warmed=True):
    from egoprior import _lib, synth
    from egoprior import engine as eng
    L, dev = engine.L, engine.device
    d, rgb, m = [t[0].to(dev).contiguous() for t in synth.make_frames(H, W, [frame_id], 8)]
    T = np.linalg.inv(synth.exo_w2c())
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = [0.1, -0.2, 0.3]
    p = eng.make_unproject_params(H, W, s, select, synth.exo_intrinsics(H, W), c2w, T)
    ns = ((H + s - 1) // s) * ((W + s - 1) // s)
    res = []
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for which in ("workspace", "noalloc"):
        pts = torch.full((ns, 4), -7.0, dtype=torch.float32, device=dev)
        x64 = torch.full((ns, 3), -7.0, dtype=torch.float64, device=dev)
        cnt = torch.full((1,), -1, dtype=torch.int32, device=dev)
        flg = torch.full((1,), -1, dtype=torch.int32, device=dev)
        if which == "workspace":
            ws = torch.empty(L.egoprior_unproject_workspace_bytes(H, W, s), dtype=torch.uint8, device=dev)
            _lib.check(L.egoprior_unproject_frame(C.byref(p), _p(d), _p(rgb), _p(m), _p(pts), _p(x64), _p(cnt), _p(flg),
                                                  _p(ws), ws.numel(), stream))
        else:
            # nothing may be allocated by the call: neither through torch's allocator nor behind its back (hipMemGetInfo)
            torch.cuda.synchronize()
            before = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev), torch.cuda.mem_get_info(dev)[0])
            _lib.check(L.egoprior_unproject_frame_noalloc(C.byref(p), _p(d), _p(rgb), _p(m), _p(pts), _p(x64), _p(cnt),
                                                          _p(flg), stream))
            torch.cuda.synchronize()
            after = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev), torch.cuda.mem_get_info(dev)[0])
            if warmed:  # (the very first call on a device may load the code object that holds the scratch array)
                assert after == before, f"device memory changed across the call: {before} -> {after}"
        torch.cuda.synchronize()
        res.append((pts, x64, cnt, flg))
    return res


def test_noalloc_unprojection_matches_the_workspace_entry_point(engine):
    from egoprior import _lib
    # first call: a 1080p frame at subsample 1 (2 025 compaction blocks); second: a smaller frame, other rule and stride
    for (H, W, s, select, fid) in ((1080, 1920, 1, _lib.SELECT_STATIC, 2), (270, 484, 2, _lib.SELECT_DYNAMIC, 5),
                                   (120, 208, 1, _lib.SELECT_STATIC, 1)):
        (pa, xa, ca, fa), (pb, xb, cb, fb) = _both(engine, H, W, s, select, fid, warmed=(H != 1080))
        n = int(ca.item())
        assert n == int(cb.item()) and n > 0
        assert int(fa.item()) == int(fb.item())
        # every byte, including the untouched tail of the output buffers
        assert torch.equal(pa.view(torch.int32), pb.view(torch.int32))
        assert torch.equal(xa.view(torch.int64), xb.view(torch.int64))


def test_noalloc_refuses_frames_beyond_its_scratch(engine):
    from egoprior import _lib
    from egoprior import engine as eng
    H, W = 4096, 4096  # 16.8 M samples at subsample 1: more than the static scratch covers
    p = eng.make_unproject_params(H, W, 1, _lib.SELECT_STATIC, [1, 1, 0, 0], np.eye(4), np.eye(4))
    dummy = torch.zeros(64, dtype=torch.uint8, device=engine.device)
    rc = engine.L.egoprior_unproject_frame_noalloc(C.byref(p), _p(dummy), _p(dummy), None, _p(dummy), None, _p(dummy),
                                                   _p(dummy), None)  # the size check comes before any launch
    assert rc == _lib.E_UNSUPPORTED
    assert b"exceed the static scratch" in engine.L.egoprior_last_error()
    p2 = eng.make_unproject_params(H, W, 2, _lib.SELECT_STATIC, [1, 1, 0, 0], np.eye(4), np.eye(4))
    assert ((H + 1) // 2) * ((W + 1) // 2) <= _lib.NOALLOC_MAX_SAMPLES  # ... while subsample 2 would fit
