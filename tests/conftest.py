import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "egox-egopriorrenderer_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from egoprior.engine import Engine
    return Engine("cuda:0")


@pytest.fixture(scope="session")
def check_build(engine):
    """(engine, library) of the -DEP_CHECK build of the HIP library: the product kernels plus a checksum that every slice
    of the splat kernel stores behind the per-pixel lists it hands over and that the merging wave verifies; skips when the
    build is not there (make -C egox-egopriorrenderer_amd/csrc libegoprior_hip_check.so)."""
    import copy
    import ctypes as C
    from egoprior import _lib
    path = _lib.CSRC / "libegoprior_hip_check.so"
    if not path.exists():
        pytest.skip("libegoprior_hip_check.so is not built")
    L = C.CDLL(str(path))
    for name, (rt, args) in _lib.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = rt, args
    L.egoprior_debug_check_counts.restype = C.c_int
    L.egoprior_debug_check_counts.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]
    E2 = copy.copy(engine)
    E2.L = L
    E2._ws = {}

    def counts(reset=True):
        seen, failed = C.c_ulonglong(0), C.c_ulonglong(0)
        assert L.egoprior_debug_check_counts(C.byref(seen), C.byref(failed), int(reset)) == 0
        return seen.value, failed.value
    E2.check_counts = counts
    return E2, L
