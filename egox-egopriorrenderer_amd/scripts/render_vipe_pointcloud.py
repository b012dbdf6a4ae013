#!/usr/bin/env python3
"""CLI drop-in for the reference's scripts/render_vipe_pointcloud.py (same flags, same output path
<out_dir>/<stem(input_dir)>/ego_Prior.mp4).  Extra, opt-in: launch under torchrun to shard frames over the GPUs of a
node (see egoprior.distributed)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

if __name__ == "__main__":
    # under torchrun the frame-sharded path runs, with one rank too (its collectives then are one-rank collectives)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        from egoprior.distributed import main_distributed
        main_distributed()
    else:
        from egoprior.render_vipe_pointcloud import main
        main()
