"""CPU, world_size 2 and 8 over gloo (also with fewer frames than ranks, i.e. empty shards): the N>1 orchestration (frame sharding, mean-background all-reduce, cloud all-gather,
frame gather).  The per-rank arithmetic is done here by the oracle (this is a test: the product path runs the HIP
kernels in its place); what is under test is that sharding + the collectives reproduce the single-process result."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, tmp, F=7):
    for p in (ROOT, ROOT / "egox-egopriorrenderer_amd"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egoprior import distributed as D, synth
    from oracle import oracle as O
    cpu = torch.device("cpu")
    H, W = 24, 32
    d, rgb, m = [t.numpy() for t in synth.make_frames(H, W, range(F), F)]
    mine = D.shard_range(F, rank, world)
    # --- mean background: per-rank folded sums + counts, all-reduced (a rank without frames contributes zeros)
    if len(mine):
        md = np.stack([np.where(O.reliable_depth_mask_range(d[f]) & (m[f] == 0), d[f], np.nan) for f in mine])
        mr = np.stack([np.where((O.reliable_depth_mask_range(d[f]) & (m[f] == 0))[..., None],
                                rgb[f].astype(np.float32) / np.float32(255), np.nan) for f in mine])
        sums = np.concatenate([O.cascade_nansum(md)[None], np.moveaxis(O.cascade_nansum(mr), -1, 0)], axis=0)
        cnt = (~np.isnan(md)).sum(axis=0).astype(np.int32)
    else:
        sums, cnt = np.zeros((4, H, W), dtype=np.float32), np.zeros((H, W), dtype=np.int32)
    ts, tc = torch.from_numpy(sums.copy()), torch.from_numpy(cnt.copy())
    D.allreduce_mean_state(dist, ts, tc)
    # --- clouds: all-gather in rank order == frame order
    pts = [O.unproject_frame(d[f], rgb[f], m[f], synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32),
                             np.eye(4), 2, "static")[0].astype(np.float32) for f in mine]
    local = torch.from_numpy(np.concatenate(pts, axis=0)) if pts else torch.zeros((0, 3))
    allp, counts = D.allgather_clouds(dist, local, local.shape[0], world, cpu)
    # --- frames: gather on rank 0
    fr = torch.stack([torch.full((4, 5, 3), f, dtype=torch.uint8) for f in mine]) if len(mine) else \
        torch.zeros((0, 4, 5, 3), dtype=torch.uint8)
    allf = D.gather_frames(dist, fr, F, rank, world, cpu)
    flag = D.allreduce_max_flag(dist, 1 if (F - 1) in mine else 0, cpu)  # only the owner of the last frame raises it
    assert flag == 1
    if rank == 0:
        np.savez(Path(tmp) / "out.npz", sums=ts.numpy(), cnt=tc.numpy(), allp=allp.numpy(), allf=allf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_in_order():
    from egoprior.distributed import shard_range
    for n in (0, 1, 7, 8, 49, 300):
        for w in (1, 2, 3, 8):
            parts = [list(shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@pytest.mark.parametrize("world,F", [(2, 7), (8, 19), (8, 3), (3, 1)])
def test_gloo_ranks_match_single_process(tmp_path, world, F):
    """(8, 3) and (3, 1): fewer frames than ranks -- ranks with an empty shard take part in every collective with zero
    rows (all-gather of clouds, gather of frames) or zero sums (mean background) and the result is the single-process one."""
    port = 29500 + (os.getpid() % 2000) + 7 * world + F
    mp.spawn(_worker, args=(world, port, str(tmp_path), F), nprocs=world, join=True)
    out = np.load(tmp_path / "out.npz")
    from egoprior import synth
    from oracle import oracle as O
    H, W = 24, 32
    d, rgb, m = [t.numpy() for t in synth.make_frames(H, W, range(F), F)]
    ok = np.stack([O.reliable_depth_mask_range(d[f]) & (m[f] == 0) for f in range(F)])
    assert np.array_equal(out["cnt"], ok.sum(axis=0))            # integer counts: exact for any rank count
    full = O.cascade_nansum(np.where(ok, d, np.nan))
    np.testing.assert_allclose(out["sums"][0], full, rtol=1e-6)  # float sums: partials folded per rank
    pts = [O.unproject_frame(d[f], rgb[f], m[f], synth.exo_intrinsics(H, W), np.eye(4, dtype=np.float32), np.eye(4), 2,
                             "static")[0].astype(np.float32) for f in range(F)]
    assert np.array_equal(out["allp"], np.concatenate(pts, axis=0))  # same order as one process
    assert [int(x) for x in out["allf"][:, 0, 0, 0]] == list(range(F))
