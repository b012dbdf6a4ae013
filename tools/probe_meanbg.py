"""Diagnostic: time of the mean-background accumulation over F resident 1080p frames."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "egox-egopriorrenderer_amd")]
import torch
from egoprior import _lib, engine as eng, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 100
E = eng.Engine("cuda:0")
H, W = 1080, 1920
d, rgb, m = synth.make_frames(H, W, list(range(F)), F, device="cuda:0")
for rep in range(3):
    state = E.meanbg_new_state(H, W)
    torch.cuda.synchronize()
    _lib.profile_begin()
    E.meanbg_accumulate(state, d, rgb, m, 0)
    k = _lib.profile_end()
    for name, (ms, calls) in k.items():
        us = 1e3 * ms / F
        print(f"{name}: {ms:.3f} ms for {F} frames = {us:.2f} us per frame, {8 * H * W / us / 1e3:.0f} GB/s "
              f"({8 * H * W / us / 8e6 * 100:.1f}% of 8 TB/s)")
