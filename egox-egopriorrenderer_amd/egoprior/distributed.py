"""Frame-sharded rendering over the GPUs of one node: one process per GPU (torchrun), torch.distributed with the
"nccl" backend (RCCL over xGMI on ROCm).  The reference has no counterpart: its renderer is one process on one GPU.

What is exchanged:
  * --use_mean_bg: every rank accumulates the per-pixel sums (depth, r, g, b) and counts of ITS frames, then ONE
    all-reduce(sum) of the 4 float32 planes and one of the int32 count plane make every rank hold the same totals;
    every rank then finalises the identical static cloud.  (Partial sums are folded per rank, so the float32 depth/colour
    means can differ from a single-GPU run in the last bit; the exact integer counts cannot.)
  * accumulated background (default mode): every rank unprojects its frames' static points; the per-rank clouds are
    all-gathered in rank (= frame) order, so point indices match the single-GPU cloud.
  * finished uint8 frames are gathered on rank 0, which writes the video.
Frames are split into contiguous blocks: rank r renders frames [r*F/G, (r+1)*F/G).
"""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import List, Sequence

import numpy as np

logger = logging.getLogger(__name__)


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block of rank `rank` when n_items are split as evenly as possible in order."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def _on_host(dist, group) -> bool:
    """gloo moves host tensors only; RCCL ("nccl") moves device tensors."""
    return dist.get_backend(group) == "gloo"


def allreduce_mean_state(dist, sums, counts, group=None):
    """In-place all-reduce(sum) of the folded mean-background sums (4,H,W) float32 and counts (H,W) int32."""
    if _on_host(dist, group) and sums.is_cuda:
        hs, hc = sums.cpu(), counts.cpu()
        dist.all_reduce(hs, group=group)
        dist.all_reduce(hc, group=group)
        sums.copy_(hs)
        counts.copy_(hc)
        return sums, counts
    dist.all_reduce(sums, group=group)
    dist.all_reduce(counts, group=group)
    return sums, counts


def allreduce_max_flag(dist, flag: int, device, group=None) -> int:
    """The largest of the ranks' flags (the "some colour byte exceeds 1" word of a cloud gathered from all ranks)."""
    import torch
    t = torch.tensor([int(flag)], dtype=torch.int32, device="cpu" if _on_host(dist, group) else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def gather_frames(dist, local_frames, n_total: int, rank: int, world: int, device, group=None):
    """Rank 0 receives every rank's (n_r, H, W, 3) uint8 block in rank order; returns the (n_total, H, W, 3) tensor on
    rank 0 and None elsewhere.  Blocks may differ in length by one frame: they are padded to the longest."""
    import torch
    n_max = max(len(shard_range(n_total, r, world)) for r in range(world))
    shape = tuple(local_frames.shape[1:])
    pad = torch.zeros((n_max,) + shape, dtype=torch.uint8, device=device)
    pad[:local_frames.shape[0]] = local_frames
    if _on_host(dist, group):
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([bufs[r][:len(shard_range(n_total, r, world))] for r in range(world)], dim=0).to(device)


def allgather_clouds(dist, points, count: int, world: int, device, group=None):
    """All ranks receive the concatenation, in rank order, of every rank's first `count` rows of `points` (n,4)."""
    import torch
    host = _on_host(dist, group)
    cdev = "cpu" if host else device
    counts = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([count], dtype=torch.int64, device=cdev), group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1)
    pad = torch.zeros((n_max, points.shape[1]), dtype=points.dtype, device=device)
    pad[:count] = points[:count]
    if host:  # all_gather of device tensors is an RCCL matter; gloo takes the host path
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0).to(device), counts


class DataGroup:
    """The process group that carries the data collectives, alive only inside `with`.  The control plane (barriers,
    small flags) is the default gloo group on the host.  With an RCCL communicator alive, kernels issued on different HIP
    streams overlap far less on this stack (measured with one rank and no collective in flight: 5 700 -> 4 800 frames/s),
    so the communicator exists around the all-reduce / all-gather of the set-up and around the final gather, and not
    while the frames render.  backend "gloo" (several ranks on one GPU, a rehearsal): the default group itself."""

    def __init__(self, dist, backend: str, device):
        self.dist, self.backend, self.device, self.group = dist, backend, device, None

    def __enter__(self):
        if self.backend == "nccl":
            self.group = self.dist.new_group(backend="nccl", device_id=self.device)
        return self.group  # None = default (gloo) group

    def __exit__(self, *exc):
        if self.group is not None:
            import torch
            torch.cuda.synchronize(self.device)
            self.dist.destroy_process_group(self.group)
            self.group = None
        return False


def main_distributed(argv=None):
    """The CLI under torchrun: same flags and output as the single-GPU main()."""
    import torch
    import torch.distributed as dist

    from . import _lib
    from . import engine as eng
    from . import render_vipe_pointcloud as rv
    from .io import SKIPPED, write_video

    logging.basicConfig(level=logging.INFO)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args = rv.get_parser().parse_args(argv)
    if args.dist_backend == "gloo":  # ranks may share GPUs
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the host name need not resolve
    dist.init_process_group(backend="gloo")  # control plane; the data collectives get a group of their own (DataGroup)
    data = DataGroup(dist, args.dist_backend, device)
    args.out_dir = rv.configure_output_directory(args)
    if rank == 0:
        os.makedirs(args.out_dir, exist_ok=True)
    _, exo_extrinsic, ego_K, ego_list = rv.load_camera_params_from_meta(args.meta_json_path, args.input_dir)
    T = np.linalg.inv(exo_extrinsic)
    rv.validate_frame_range(args, len(ego_list), args.input_dir)
    E = eng.Engine(device)
    rv._ENGINE = E
    artifact = rv._first_artifact(args.input_dir)
    from .io import read_pose_artifacts
    n_frames = len(read_pose_artifacts(artifact.pose_path)[0])
    mine = shard_range(n_frames, rank, world)
    to_render = list(range(args.start_frame, args.end_frame + 1))
    my_slots = shard_range(len(to_render), rank, world)
    # this rank decodes the colour images, depth maps and masks of its own frames only (plus frame 0, whose size,
    # intrinsics and camera type every rank needs)
    need = set(mine) | {slot for slot in my_slots if slot < n_frames} | {0}
    # (c2w, rgb, intr, type, depth, mask) per frame; rgb / depth / mask = None or io.SKIPPED for other ranks' frames.  The
    # rank's share of the clip is decoded once into HBM where the containers allow it (egoprior.clip), else streamed.
    from .clip import DeviceClip
    clip = None if os.environ.get("EGOPRIOR_STREAM_CLIP") == "1" else DeviceClip.load(E, artifact, wanted=need)
    frames = list(clip.frames(tolerant=True) if clip is not None else rv._frames(artifact, tolerant=True, wanted=need))
    n_frames = len(frames)
    s = 2
    # ---- static cloud ---------------------------------------------------------------------------------------------
    # Frame 0 (decoded by every rank) supplies the frame size and the intrinsics / camera type that serve every frame of
    # the accumulated cloud: the reference builds its ray table once, from the first frame that has a depth map, and a
    # depth archive either covers frame 0 or is missing altogether (then there is no cloud to build).
    first = frames[0]
    H, W = first[1].shape[:2]
    tables = rv._RayTables()  # device ray tables of non-pinhole exo cameras (None for PINHOLE)
    if args.use_mean_bg:
        state = E.meanbg_new_state(H, W)
        k = 0
        for f in mine:
            c2w, rgb, intr, ct, depth, mask = frames[f]
            if depth is None or depth is SKIPPED:
                continue
            E.meanbg_accumulate(state, depth, rgb, mask, k)
            k += 1
        sums, cnt = E.meanbg_export(state)
        torch.cuda.synchronize(device)
        with data as g:
            allreduce_mean_state(dist, sums, cnt, group=g)
        E.meanbg_import(state, sums, cnt)
        md, mc = E.meanbg_finalize(state)
        bg = E.unproject_frame(md, mc, None, first[2], first[0], T, subsample=s, select=_lib.SELECT_ALL,
                               rays=tables.get(first[3], first[2], H, W, s))
    else:
        parts, flag = [], 0
        for f in mine:
            c2w, rgb, intr, ct, depth, mask = frames[f]
            if depth is None or depth is SKIPPED:
                continue
            c = E.unproject_frame(depth, rgb, mask, first[2], c2w, T, subsample=s, select=_lib.SELECT_STATIC,
                                  rays=tables.get(first[3], first[2], H, W, s))  # one table, as the reference builds it once
            parts.append(c.points[:c.n()].clone())
            flag = max(flag, int(c.flags.item()))
        local = torch.cat(parts, dim=0) if parts else torch.zeros((0, 4), device=device)
        torch.cuda.synchronize(device)
        with data as g:
            allpts, _ = allgather_clouds(dist, local, local.shape[0], world, device, group=g)
            flag = allreduce_max_flag(dist, flag, device, group=g)  # the `cols.max() > 1` rule on the whole cloud
        bg = E.new_cloud(allpts.shape[0])
        bg.points[:allpts.shape[0]] = allpts
        bg.count.fill_(allpts.shape[0])
        bg.flags.fill_(flag)
    # ---- my frames (no communicator exists while they render) ------------------------------------------------------
    size = (448, 448)
    rad = tan = thin = foc = pp = None
    if args.fish_eye_rendering:
        rad, tan, thin, foc, pp = rv.load_aria_distortion_coeffs(args.online_calibration_path, frame_idx=0)
    # this rank's block of frames goes through the same pipelined frame loop as the single-GPU CLI
    items = None
    if not args.only_bg:
        items = iter([frames[slot] if slot < n_frames else None for slot in my_slots])
    imgs = rv._render_sequence(bg, items, [ego_list[to_render[slot]] for slot in my_slots], ego_K, size,
                               args.point_size, args.fish_eye_rendering, (rad, tan, thin, foc, pp), (2880, 2880), T)
    out = [torch.from_numpy(np.ascontiguousarray(img)).to(device) for img in imgs]
    local = torch.stack(out) if out else torch.zeros((0, size[0], size[1], 3), dtype=torch.uint8, device=device)
    torch.cuda.synchronize(device)
    with data as g:
        allf = gather_frames(dist, local, len(to_render), rank, world, device, group=g)
    if rank == 0:
        write_video(Path(args.out_dir) / "ego_Prior.mp4", list(allf.cpu().numpy()), fps=30)
    dist.barrier()
    dist.destroy_process_group()
