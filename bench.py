#!/usr/bin/env python3
"""Benchmark of the ego-prior rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic ViPE-format clip, 1920x1080 exo depth + RGB + instance mask per frame
(spatial_subsample 1, ~2 M candidate points per frame), static background = per-pixel NaN-aware mean over the clip,
rendered to a 480x832 pinhole ego camera with points_per_pixel = 20.  One step = one ego frame of the reference's
frame loop (scripts/render_vipe_pointcloud.py:1178-1232) with every input already resident in HBM:
    dynamic points of exo frame f (5x5 reliability mask, instance rule, unprojection, float64 exo->world)
    + projection of [static cloud | dynamic points] + tile binning + exact top-20 splat + compositor + uint8 frame.
The static cloud is built once per clip before the timed region (as the reference does) and its time is reported
separately.  With N > 1 every rank renders its own K frames (weak scaling, frames are independent); the mean-background
sums are all-reduced over RCCL during set-up; the finished frames are gathered on rank 0 over RCCL right after the
timed region and that gather is timed on its own (`gather_ms`), because an RCCL communicator that is alive while the
frames render costs about a sixth of the rendering rate on this stack (`--gather blocks` keeps it inside).  Barriers
and the max over ranks go through a gloo group on the host.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "egox-egopriorrenderer_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--point-size", type=float, default=5.0)  # scripts/render_vipe.sh:13
    ap.add_argument("--exo", type=str, default="1080x1920")
    ap.add_argument("--ego", type=str, default="480x832")
    ap.add_argument("--frames", type=int, default=300, help="exo frames per rank resident in HBM")
    ap.add_argument("--subsample", type=int, default=1)
    ap.add_argument("--skip-cpu", action="store_true", help="skip the host-CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (egoprior_set_option)")
    ap.add_argument("--gather", choices=["after", "blocks", "end", "none"], default="after",
                    help="N > 1: how finished frames reach rank 0.  after (default): one RCCL gather after the timed "
                         "region, timed on its own (gather_ms); the RCCL communicator does not exist while the frames "
                         "render.  blocks / end: gathered inside the timed region, block by block behind the rendering "
                         "or once at its end.  none: no process group at all")
    ap.add_argument("--stream-pick", type=str, default="",
                    help="experiment: comma-separated indices into 8 freshly created torch streams to use for the "
                         "pipeline's slots 1..")
    ap.add_argument("--debug-timing", action="store_true", help="print the phases of the timed region to stderr")
    ap.add_argument("--no-dynamic", action="store_true",
                    help="experiment: render the static cloud only (skips the per-frame dynamic-point kernels)")
    ap.add_argument("--depth-f16", action="store_true",
                    help="keep the exo depth maps in HBM as float16 (the precision of the EXR artifacts) instead of float32")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N > 1 control flow on a box with ONE GPU: every rank uses cuda:0 and the data "
                         "collectives go through gloo (RCCL refuses two ranks on one device); the numbers mean nothing")
    ap.add_argument("--from-host", action="store_true",
                    help="the PCIe-inclusive rate: every frame's depth, RGB and mask start in pinned host memory and are "
                         "copied to the GPU on the frame's own stream inside the timed region (never the headline value)")
    ap.add_argument("--fisheye", action="store_true",
                    help="BASELINE configs[3] instead of configs[1]: fisheye ego camera (the preset Ego-Exo4D radial / "
                         "tangential / thin-prism coefficients), not rotated; combine with --point-size 10 or 20")
    ap.add_argument("--streams", type=int, default=4,
                    help="frames in flight: consecutive frames go round-robin to this many HIP streams, each with its "
                         "own dynamic-point buffer and workspaces (frames are independent of one another)")
    return ap.parse_args()


def main():
    args = parse()
    if args.debug_timing:  # a hung run names the line it hangs on
        import faulthandler
        faulthandler.dump_traceback_later(40, exit=True)
    # Only the JSON line may reach stdout: RCCL prints a version banner there.  Everything written to file descriptor 1
    # until the end goes to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run(args)
    finally:
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        os.close(json_fd)
    if line is not None:
        print(line, flush=True)


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # under torchrun the collectives run even with one rank (a rehearsal of the N > 1 path), unless --gather none
    if world > 1 or ("TORCHELASTIC_RUN_ID" in os.environ and args.gather != "none"):
        import torch.distributed as dist_mod
        dist = dist_mod
        # control plane (barriers, max of the times, work counters): gloo on the host.  Data collectives: an RCCL group
        # that exists only while it is needed -- with an RCCL communicator alive, kernels of different HIP streams
        # overlap far less on this stack (measured: 5700 -> 4800 frames/s with one rank and no collective in flight).
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the container's host name need not resolve
        dist.init_process_group(backend="gloo")

    def rccl_group():
        if args.rehearse_one_gpu:
            return dist.new_group(backend="gloo")
        return dist.new_group(backend="nccl", device_id=dev)

    def gather_to_rank0(t, bufs, group, async_op=False):
        """dist.gather of a block of finished frames; in the one-GPU rehearsal through host copies (gloo has no gather
        for device tensors)."""
        if not args.rehearse_one_gpu:
            return dist.gather(t, bufs, dst=0, group=group, async_op=async_op)
        host = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)] if rank == 0 else None
        dist.gather(t.cpu(), host, dst=0, group=group)
        if rank == 0:
            for b, h in zip(bufs, host):
                b.copy_(h)

        class _Done:
            def wait(self):
                return True
        return _Done()

    from egoprior import _lib, synth
    from egoprior import engine as eng

    E = eng.Engine(dev)
    L = E.L
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    H, W = [int(v) for v in args.exo.split("x")]
    He, We = [int(v) for v in args.ego.split("x")]
    s = args.subsample
    Fl = args.frames
    F_total = Fl * world
    frame_ids = list(range(rank * Fl, (rank + 1) * Fl))

    # ---- synthetic clip straight into HBM --------------------------------------------------------------------------
    t0 = time.time()
    depth = torch.empty((Fl, H, W), dtype=torch.float16 if args.depth_f16 else torch.float32, device=dev)
    rgb = torch.empty((Fl, H, W, 3), dtype=torch.uint8, device=dev)
    mask = torch.empty((Fl, H, W), dtype=torch.uint8, device=dev)
    for c0 in range(0, Fl, 25):
        ids = frame_ids[c0:c0 + 25]
        d, c, m = synth.make_frames(H, W, ids, F_total, device=dev)
        depth[c0:c0 + len(ids)], rgb[c0:c0 + len(ids)], mask[c0:c0 + len(ids)] = d, c, m
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    exo_E = synth.exo_w2c()
    T_c2w = np.linalg.inv(exo_E)  # scripts/render_vipe_pointcloud.py:1368 (float32)
    intr = synth.exo_intrinsics(H, W)
    c2w = np.eye(4, dtype=np.float32)
    egos = synth.ego_orbit_w2c(F_total, exo_E)
    K_ego = synth.ego_pinhole_K(He, We)

    # ---- static background: mean over the clip (all-reduced across ranks) ------------------------------------------
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    state = E.meanbg_new_state(H, W)
    E.meanbg_accumulate(state, depth, rgb, mask, 0)
    data_group = None
    if dist is not None:
        data_group = rccl_group()
        sums, cnt = E.meanbg_export(state)
        dist.all_reduce(sums, group=data_group)
        dist.all_reduce(cnt, group=data_group)
        E.meanbg_import(state, sums, cnt)
    mean_d, mean_c = E.meanbg_finalize(state)
    bg = E.unproject_frame(mean_d, mean_c, None, intr, c2w, T_c2w, subsample=s, select=_lib.SELECT_ALL)
    torch.cuda.synchronize()
    t_bg = time.time() - t0
    n_bg = bg.n()
    del state

    # ---- per-frame arguments, prepared once -------------------------------------------------------------------------
    ns = ((H + s - 1) // s) * ((W + s - 1) // s)
    up = eng.make_unproject_params(H, W, s, _lib.SELECT_DYNAMIC, intr, c2w, T_c2w)
    rp = eng.make_raster_params(args.point_size, He, We, K=20, rotate_cw=not args.fisheye, colors_gt1=-1)
    cams = []
    if args.fisheye:
        from egoprior.render_vipe_pointcloud import load_aria_distortion_coeffs
        radial, tangential, thin_prism, _, _ = load_aria_distortion_coeffs(None)
    for f in frame_ids:
        R, t = eng.split_w2c(egos[f])
        if args.fisheye:  # scripts/render_vipe_pointcloud.py:1040-1062 with the ego intrinsics of the meta file
            cams.append(eng.fisheye_camera(R, t, K_ego[0, 0], K_ego[0, 2], K_ego[1, 2], radial, tangential, thin_prism,
                                           He, We))
        else:
            cams.append(eng.pinhole_camera(R, t, K_ego, He, We))
    CH = 64  # frames per gather
    ring = min(-(-max(args.steps, 1) // CH) * CH, 16 * CH)
    frames_out = torch.empty((ring, He, We, 3) if args.fisheye else (ring, We, He, 3), dtype=torch.uint8, device=dev)
    # frames are independent: consecutive ones go round-robin to `--streams` HIP streams (egoprior.engine.FramePipeline)
    extra = None
    if args.stream_pick:
        pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
        extra = [pool[int(v)] for v in args.stream_pick.split(",")]
        args.streams = len(extra) + 1
    pipe = eng.FramePipeline(E, bg, rp, He, We, exo_hw=(H, W), subsample=s, slots=args.streams, extra_streams=extra,
                             oversubscribe=args.streams > eng.hardware_queues())  # an explicit --streams N is an experiment
    dyn = pipe.dyns[0]

    def step_inputs(i, slot=None):
        f = i % Fl
        return {} if args.no_dynamic else dict(up=up, depth=depth[f], rgb=rgb[f], mask=mask[f])

    if args.from_host:
        # a window of the clip in pinned host memory and one set of device staging buffers per slot; the copies are
        # queued on the slot's stream, so they are ordered with that slot's kernels and overlap the other slots' work
        n_host = min(Fl, 64)
        h_depth, h_rgb, h_mask = (t[:n_host].cpu().pin_memory() for t in (depth, rgb, mask))
        stage = [(torch.empty_like(depth[0]), torch.empty_like(rgb[0]), torch.empty_like(mask[0]))
                 for _ in range(pipe.n_slots)]
        host_step = [0]

        def step_inputs(i, slot=None):  # noqa: F811  (replaces the resident-clip version above)
            f = i % n_host
            k = host_step[0] % pipe.n_slots if slot is None else slot
            host_step[0] += 1
            with torch.cuda.stream(pipe.streams[k]):
                for dst, src in zip(stage[k], (h_depth[f], h_rgb[f], h_mask[f])):
                    dst.copy_(src, non_blocking=True)
            return dict(up=up, depth=stage[k][0], rgb=stage[k][1], mask=stage[k][2], slot=k)

    def step(i, slot=None):
        kw = step_inputs(i, slot)
        if slot is not None:
            kw["slot"] = slot
        pipe.submit(cams[i % Fl], frames_out[i % ring], **kw)

    # N > 1: every finished block of CH frames is gathered on rank 0 while the next block renders (the collective runs
    # on RCCL's own stream, ordered after the block by pipe.join()); all K frames of every rank arrive within the timed
    # region.
    gather_bufs = None
    if dist is not None and rank == 0:
        gather_bufs = [[torch.empty((CH, We, He, 3), dtype=torch.uint8, device=dev) for _ in range(world)]
                       for _ in range(2)]

    for i in range(args.warmup):
        step(i)
    pipe.join()
    if dist is not None and args.gather in ("blocks", "end"):
        # the first gather sets up RCCL's point-to-point channels: keep that out of the timed region
        gather_to_rank0(frames_out[0:CH], gather_bufs[0] if rank == 0 else None, data_group)
    torch.cuda.synchronize()
    if dist is not None and args.gather == "after":
        dist.destroy_process_group(data_group)  # no RCCL communicator while the frames render
        data_group = None
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    works = []
    for i in range(args.steps):
        # output slot: position in the current block of CH frames, blocks laid out one after the other
        pipe.submit(cams[(args.warmup + i) % Fl], frames_out[i % ring], **step_inputs(args.warmup + i))
        if dist is not None and args.gather == "blocks" and ((i + 1) % CH == 0 or i + 1 == args.steps):
            pipe.join()
            blk = i // CH
            lo = (blk * CH) % ring
            if len(works) >= 2:
                works[-2].wait()  # its receive buffers are used again now
            works.append(gather_to_rank0(frames_out[lo:lo + CH], gather_bufs[blk % 2] if rank == 0 else None, data_group, async_op=True))
    t_issue = time.perf_counter() - t0  # host time to queue all steps (the GPU is still working)
    pipe.join()
    if dist is not None and args.gather == "end":
        for lo in range(0, min(args.steps, ring), CH):
            gather_to_rank0(frames_out[lo:lo + CH], gather_bufs[0] if rank == 0 else None, data_group)
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    t_done = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if args.debug_timing:
        print(f"[rank {rank}] issue {1e3 * t_issue:.2f} ms, frames+gathers done {1e3 * t_done:.2f} ms, "
              f"after barrier {1e3 * elapsed:.2f} ms", file=sys.stderr)
    gather_ms = None
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if args.gather == "after":
            # the K frames of every rank go to rank 0 over RCCL now; timed separately and reported as gather_ms
            data_group = rccl_group()
            gather_to_rank0(frames_out[0:CH], gather_bufs[0] if rank == 0 else None, data_group)  # set-up
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            for lo in range(0, min(args.steps, ring), CH):
                gather_to_rank0(frames_out[lo:lo + CH], gather_bufs[0] if rank == 0 else None, data_group)
            torch.cuda.synchronize()
            dist.barrier()
            gather_ms = 1e3 * (time.perf_counter() - tg)
    assert int(pipe.status.abs().sum().item()) == 0, "tile lists overflowed"

    # ---- work accounting (outside the timed region) -----------------------------------------------------------------
    pts_entering = 0
    pts_candidates = 0
    n_dyn_sum = 0
    used = sorted(set((args.warmup + i) % Fl for i in range(args.steps)))
    per_frame = {}
    for f in used:
        E.unproject_frame(depth[f], rgb[f], mask[f], intr, c2w, T_c2w, subsample=s, select=_lib.SELECT_DYNAMIC, out=dyn)
        nd = dyn.n()
        ndc = E.project_points(cams[f], bg, dyn)[:n_bg + nd]
        per_frame[f] = (int((ndc[:, 2] >= 0).sum().item()), nd)
    for i in range(args.steps):
        f = (args.warmup + i) % Fl
        pts_entering += per_frame[f][0]
        n_dyn_sum += per_frame[f][1]
        pts_candidates += n_bg + ns
    if dist is not None:
        tt = torch.tensor([pts_entering, pts_candidates, n_dyn_sum], dtype=torch.int64)
        dist.all_reduce(tt)
        pts_entering, pts_candidates, n_dyn_sum = [int(v) for v in tt.tolist()]

    # ---- per-kernel timing with HIP events (separate pass, same frames) ---------------------------------------------
    kern = {}
    if rank == 0:
        n_prof = args.steps  # the same frames as the timed region (their cost varies along the ego orbit)
        torch.cuda.synchronize()
        _lib.profile_begin()  # every kernel bracketed by HIP events; one stream, so nothing overlaps the timed kernel
        for i in range(n_prof):
            step(args.warmup + i, 0)
        kern = _lib.profile_end()
        torch.cuda.synchronize()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return None

    n_frames_total = args.steps * world
    cam_kind = "fisheye" if args.fisheye else "pinhole"
    fps = n_frames_total / elapsed
    n_avg = pts_entering / n_frames_total  # points entering the binning stage per frame
    n_in_avg = n_bg + n_dyn_sum / n_frames_total
    alg_bytes = {
        "k_reliable_flags": (2 if args.depth_f16 else 4) * H * W + 2 * ns,
        "k_flag_block_count": ns,
        "k_block_scan": 0,
        "k_unproject_scatter": ns + (n_dyn_sum / n_frames_total) * (16 + 7),
        "k_project_count": 32 * n_in_avg,
        "k_tile_scan": 0,
        "k_tile_fill": 16 * n_in_avg,
        "k_tile_fill_pieces": 16 * n_in_avg,
        "k_splat_sweep": 16 * n_avg + 3 * He * We,
        "k_splat_bubble": 16 * n_avg + 3 * He * We,
    }
    kernels = {}
    for name, (ms, calls) in kern.items():
        avg_us = 1e3 * ms / max(calls, 1)
        b = alg_bytes.get(name, 0)
        gbs = (b / (avg_us * 1e-6)) / 1e9 if avg_us > 0 else 0.0
        kernels[name] = {"avg_us": round(avg_us, 2), "calls": calls, "algorithmic_bytes": int(b),
                         "GBps": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
    dominant = max(kernels, key=lambda k: kernels[k]["avg_us"] * kernels[k]["calls"]) if kernels else None
    roofline = None
    if dominant:
        k = kernels[dominant]
        # HBM figures of the dominant kernel (algorithmic bytes over its measured duration) ...
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": k["frac_hbm"], "traffic": None, "traffic_ratio": None, "avg_us": k["avg_us"]}
        tr = ROOT / "profiles" / "pmc_traffic.json"
        pmc = {}
        if tr.exists():
            try:
                pmc = json.load(open(tr))
            except Exception:
                pmc = {}
        rec = pmc.get("kernels", {}).get(dominant)
        if rec:
            # Counters come from the committed rocprofv3 passes (profiles/), collected on `pmc["_command"]`; the launch
            # duration is measured here.  `frames_match` says whether this run renders the frames the counters cover (the
            # splat's cost follows the ego orbit, so figures from different frames do not combine exactly).
            same = (args.steps == pmc.get("_steps") and args.warmup == pmc.get("_warmup") and args.frames == 300
                    and args.point_size == 5.0 and not args.fisheye and args.subsample == 1)
            roofline["traffic"] = rec["bytes_corrected"]
            roofline["traffic_uncorrected"] = rec["bytes_uncorrected"]
            if k["algorithmic_bytes"]:
                roofline["traffic_ratio"] = round(rec["bytes_corrected"] / k["algorithmic_bytes"], 2)
                roofline["traffic_ratio_uncorrected"] = round(rec["bytes_uncorrected"] / k["algorithmic_bytes"], 2)
            roofline["source"] = (f"profiles/pmc_traffic.json ({pmc.get('_round')}: {pmc.get('_command')}; "
                                  f"{pmc.get('_frames')}); avg_us measured here with HIP events on this run's frames "
                                  f"{args.warmup}..{args.warmup + args.steps - 1}")
            roofline["frames_match"] = bool(same)
            sq, ldsc = rec.get("sq", {}), rec.get("lds", {})
            insts = sq.get("SQ_INSTS_VALU")
            if insts and dominant in ("k_splat_sweep", "k_splat_bubble"):
                # What bounds the splat kernel is vector-instruction issue (exact top-20 per pixel: hit tests, sort, merge),
                # so the fraction is taken against that peak: one wave64 VALU instruction per 2 cycles and SIMD (the rate
                # MI355X_MICROARCH.md gives for a SIMD shared by two or more waves), 1024 SIMDs, 2.4 GHz.  The HBM figures
                # stay alongside.  `valu_busy_measured` is the VALU pipes' own busy counter over the kernel's duration: the
                # instruction mix here (64-bit compares, DPP operands, packed float math, three-operand forms) issues at 3
                # to 4 cycles per instruction, not 2 (tools/ubench/valu_issue.hip measures the forms one by one).
                simds, clock = 1024, 2.4e9
                peak = simds * clock / 2 / 1e9
                ach = insts / (k["avg_us"] * 1e-6) / 1e9
                hbm = {"achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["frac_hbm"],
                       "algorithmic_bytes": k["algorithmic_bytes"]}
                roofline.update({"bound": "valu", "achieved": round(ach, 1), "peak": round(peak, 1),
                                 "unit": "G wave-instructions/s", "frac": round(ach / peak, 3),
                                 "valu_wave_insts_per_launch": insts, "valu_cycles_per_inst_assumed": 2, "hbm": hbm})
                busy = sq.get("SQ_ACTIVE_INST_VALU")
                if busy:  # quad-cycles in which a SIMD's vector ALU was executing, summed over the SIMDs
                    roofline["valu_busy_measured"] = round(busy * 4 / (simds * clock * k["avg_us"] * 1e-6), 3)
                roofline["note"] = ("algorithmic bytes = 16 B per projected point + 3 B per output pixel; the measured "
                                    "traffic is a multiple of that: tile lists (8 B per point-tile pair), the gathers of NDC "
                                    "x, y for the entries that pass the threshold filter and the compositor's gathers of the "
                                    "winners' records")
            wc, ga = sq.get("SQ_WAVE_CYCLES"), sq.get("GRBM_GUI_ACTIVE")
            if wc and ga:
                # wave-cycles (units of 4 clocks) over the kernel's busy clocks = waves resident on average; the counter file
                # holds GRBM_GUI_ACTIVE summed over the 8 XCDs (it comes to 8 x duration x 2.4 GHz); the kernel runs 3
                # waves per SIMD, 3072 on the chip
                busy_clocks = ga / 8.0
                roofline["resident_waves_avg"] = round(4.0 * wc / busy_clocks, 1)
                roofline["occupancy"] = round(4.0 * wc / busy_clocks / 3072.0, 3)
            if ldsc.get("SQ_LDS_IDX_ACTIVE"):
                roofline["lds_conflict_frac"] = round(ldsc.get("SQ_LDS_BANK_CONFLICT", 0) / ldsc["SQ_LDS_IDX_ACTIVE"], 3)

    cpu = None
    if not args.skip_cpu and world == 1:
        cpu = cpu_baseline(E, bg, dyn, depth, rgb, mask, intr, c2w, T_c2w, s, cams, egos, frame_ids, K_ego, He, We,
                           args.point_size, args.cpu_seconds)

    out = {
        "metric": f"ego-prior frames/sec (1080p exo -> 480x832 {cam_kind} ego, points_per_pixel 20)",
        "value": round(fps, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "host_issue_ms_per_step": round(1e3 * t_issue / args.steps, 4),
        "gather": args.gather if dist is not None else None,
        "gather_ms": None if gather_ms is None else round(gather_ms, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{3 if args.fisheye else 1}]: {W}x{H} exo depth+RGB+mask, subsample {s}, mean-bg static cloud "
                               f"({n_bg} pts) + per-frame dynamic points -> {He}x{We} {cam_kind} ego, point_size "
                               f"{args.point_size}, K=20, {Fl} exo frames/rank resident in HBM",
                   "exo": [H, W], "ego": [He, We], "point_size": args.point_size, "subsample": s, "depth_dtype": "f16" if args.depth_f16 else "f32",
                   "frames_per_rank": Fl, "static_points": n_bg, "frames_in_flight": pipe.n_slots,
                   "inputs": "pinned host memory, copied per frame inside the timed region" if args.from_host else "resident in HBM",
                   "parallelism": f"frames sharded x{world}"},
        "gpoints_per_s": round(pts_entering / elapsed / 1e9, 4),
        "points_entering_raster_per_frame": round(n_avg, 1),
        "candidate_points_per_frame": round(pts_candidates / n_frames_total, 1),
        "bg_build_ms": round(1e3 * t_bg, 2),
        "bg_build_frames": F_total,
        "synth_gen_s": round(t_gen, 2),
        "roofline": roofline,
        "kernels": kernels,
        "cpu_baseline": cpu,
    }
    if cpu is not None and cpu.get("value"):
        out["gpu_over_cpu"] = round(fps / cpu["value"], 1)
    if dist is not None:
        dist.destroy_process_group()
    return json.dumps(out)


def cpu_baseline(E, bg, dyn, depth, rgb, mask, intr, c2w, T_c2w, s, cams, egos, frame_ids, K_ego, He, We, point_size,
                 target_s):
    """The reference's CPU path (pytorch3d naive rasterizer + compositor + numpy pre/post) restated in oracle/, timed on
    this host on ONE frame with the rasterizer restricted to a band of image rows, then extrapolated to a full frame."""
    threads = min(16, os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    threads = int(os.environ["OMP_NUM_THREADS"])
    from oracle import oracle as O
    from egoprior import _lib
    f = 0
    t0 = time.perf_counter()
    d_np, c_np, m_np = depth[f].cpu().numpy(), rgb[f].cpu().numpy(), mask[f].cpu().numpy()
    dpts, dcols = O.unproject_frame(d_np, c_np, m_np, intr, c2w, T_c2w, s, "dynamic")
    t_unproj = time.perf_counter() - t0
    bg_p, bg_c = bg.to_numpy()
    pts = np.vstack([bg_p.astype(np.float64), dpts])
    cols = np.vstack([bg_c, dcols.astype(np.uint8).reshape(-1, 3)])
    t0 = time.perf_counter()
    p32, c01 = O.filter_points(pts, cols)
    R, t = O.split_w2c(egos[frame_ids[f]])
    ndc = O.project_pinhole(p32, R, t, K_ego, He, We)
    t_pre = time.perf_counter() - t0
    r = O.radius_ndc(point_size, He, We)
    y0 = He // 2
    t0 = time.perf_counter()
    O.rasterize(ndc, r, He, We, 20, "naive", rows=(y0, y0 + threads))  # one row per thread
    t_probe = (time.perf_counter() - t0) / threads
    rows = int(max(threads, min(He - y0, target_s / max(t_probe, 1e-6))))
    rows = max(threads, rows // threads * threads)
    t0 = time.perf_counter()
    idx, zb, ds = O.rasterize(ndc, r, He, We, 20, "naive", rows=(y0, y0 + rows))
    t_band = time.perf_counter() - t0
    t0 = time.perf_counter()
    img = O.postprocess(O.composite(idx, ds, c01, r), rotate_cw=True)
    t_post = time.perf_counter() - t0
    t_frame = t_unproj + t_pre + t_band * (He / rows) + t_post
    return {"value": round(1.0 / t_frame, 6), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 ego frame, {ndc.shape[0]} points; naive per-pixel rasterizer (pytorch3d RasterizePointsNaiveCpu "
                      f"restated in C, OpenMP over rows) timed on image rows [{y0},{y0 + rows}) of {He} = "
                      f"{t_band:.2f} s, extrapolated to all rows; numpy unprojection/filter/projection "
                      f"{t_unproj + t_pre:.2f} s and compositor+post {t_post:.2f} s added as measured",
            "seconds_per_frame": round(t_frame, 3)}


if __name__ == "__main__":
    main()
