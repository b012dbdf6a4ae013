"""Dataset-scale driver: many (take, exo camera) render jobs over the GPUs of one node.

The reference does this in bash (data_preprocess/scripts/infer_vipe_all_takes.sh): `create_meta_json` (:696-840) writes
the meta.json a render needs from an Ego-Exo4D camera-pose file, `run_render` (:842-983) starts
scripts/render_vipe_pointcloud.py as a subprocess, and a pool of background PIDs (:1303-1338) keeps BATCH_SIZE of them
alive.  Here the same three pieces are Python:

  * create_meta_json ... same inputs, same selection rules, same file written; failures raise instead of `exit 1`
  * RenderJob / render_argv ... the command line run_render builds
  * run_jobs ... one long-lived worker process per GPU (the HIP library, the engine and its workspaces are set up once per
    worker instead of once per job) fed from one shared queue, longest job first, so the GPUs drain together.  A job is
    a unit: nothing is exchanged between workers, the frames of one job stay on one GPU.  (Splitting ONE clip over
    several GPUs is what egoprior.distributed is for.)

Exit codes reported per job follow the shell pool's reading (:1316-1326): 0 done, 2 skipped because an input is
missing, anything else a failure.
"""
from __future__ import annotations

import json
import logging
import os
import re
import traceback
from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable, List, Optional, Sequence

logger = logging.getLogger(__name__)

EXIT_OK, EXIT_FAILED, EXIT_SKIPPED = 0, 1, 2


# ----------------------------------------------------------------------------------------------------------------------
# meta.json
# ----------------------------------------------------------------------------------------------------------------------
def exo_camera_name(result_subdir: str) -> str:
    """`camNN` prefix of a ViPE result directory name, "cam03" when there is none (infer_vipe_all_takes.sh:709-712)."""
    m = re.match(r"^(cam[0-9]+)", result_subdir)
    return m.group(1) if m else "cam03"


def select_ego_extrinsics(ego_ext, start_frame: int, end_frame: int) -> list:
    """Per-frame ego world-to-camera matrices of frames start_frame..end_frame (inclusive) out of the `camera_extrinsics`
    entry of the aria camera, whatever shape that entry has (infer_vipe_all_takes.sh:763-789):

      dict  frame number (as a string) -> matrix: the digit keys inside the range, in numeric order; ValueError if none
      list  whose first element has 4 entries (4x4 matrices per frame -- but also the 4-float rows of one bare 3x4
            matrix, which the rule cannot tell apart): everything when the range starts at 0 and reaches the end,
            else the slice [start_frame : end_frame + 1]
      list  of exactly three other lists (one 3x4 given as rows is caught above; this is e.g. three 3x4 matrices):
            wrapped as ONE entry
      anything else: nothing
    """
    if isinstance(ego_ext, dict):
        keys = sorted(int(k) for k in ego_ext.keys() if str(k).isdigit())
        chosen = [f for f in keys if start_frame <= f <= end_frame]
        if not chosen:
            have = f"{keys[0]} ~ {keys[-1]}" if keys else "N/A ~ N/A"
            raise ValueError(f"No frames found in range {start_frame}-{end_frame}. Available: {have}")
        return [ego_ext[str(f)] for f in chosen]
    if isinstance(ego_ext, list):
        if len(ego_ext) > 0 and isinstance(ego_ext[0], list) and len(ego_ext[0]) == 4:
            if start_frame == 0 and end_frame >= len(ego_ext) - 1:
                return ego_ext
            return ego_ext[start_frame:end_frame + 1]
        return [ego_ext] if len(ego_ext) == 3 else []
    return []


def build_meta(ego_data: dict, result_subdir: str, start_frame: int, end_frame: int) -> dict:
    """The meta.json content for one exo camera of one take (infer_vipe_all_takes.sh:731-819).  ValueError names what is
    missing, in the order the reference checks it."""
    camera = exo_camera_name(result_subdir)
    if camera not in ego_data:
        available = [k for k in ego_data.keys() if k.startswith("cam")]
        raise ValueError(f"Camera '{camera}' not found in ego_pose. Available: {available}")
    exo_intrinsics = ego_data[camera].get("camera_intrinsics")
    exo_extrinsics = ego_data[camera].get("camera_extrinsics")
    aria_key = next((k for k in ego_data.keys() if k.startswith("aria")), None)  # first in file order
    if aria_key is None:
        raise ValueError("No aria camera found in ego_pose file")
    aria = ego_data[aria_key]
    ego_intrinsics = aria.get("camera_intrinsics")
    ego_list = select_ego_extrinsics(aria["camera_extrinsics"], start_frame, end_frame) \
        if "camera_extrinsics" in aria else []
    if exo_intrinsics is None:
        raise ValueError(f"exo_intrinsics not found for camera {camera}")
    if exo_extrinsics is None:
        raise ValueError(f"exo_extrinsics not found for camera {camera}")
    if ego_intrinsics is None:
        raise ValueError("ego_intrinsics not found")
    if not ego_list:
        raise ValueError(f"No ego_extrinsics found for frames {start_frame}-{end_frame}")
    # render_vipe_pointcloud matches a sample by the parent directory name of exo_path == stem of its --input_dir
    return {"test_datasets": [{
        "exo_path": f"./takes/{result_subdir}/cam01.mp4",
        "camera_intrinsics": exo_intrinsics,
        "camera_extrinsics": exo_extrinsics,
        "ego_intrinsics": ego_intrinsics,
        "ego_extrinsics": ego_list,
    }]}


def create_meta_json(ego_pose_path, take_name: str, result_subdir: str, output_path, start_frame: int = 0,
                     end_frame: int = 0) -> dict:
    """Writes <output_path> (parents created, indent 2) and returns its content.  `take_name` is carried for the log
    line only, as in the reference.  FileNotFoundError if the pose file is missing (:703-706), ValueError otherwise."""
    ego_pose_path = Path(ego_pose_path)
    if not ego_pose_path.is_file():
        raise FileNotFoundError(f"Ego camera pose not found: {ego_pose_path}")
    with open(ego_pose_path) as f:
        ego_data = json.load(f)
    meta = build_meta(ego_data, result_subdir, int(start_frame), int(end_frame))
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    with open(output_path, "w") as f:
        json.dump(meta, f, indent=2)
    logger.info("Created meta.json: %s (take %s, exo camera %s, frames %d-%d: %d ego poses)", output_path, take_name,
                exo_camera_name(result_subdir), start_frame, end_frame, len(meta["test_datasets"][0]["ego_extrinsics"]))
    return meta


# ----------------------------------------------------------------------------------------------------------------------
# jobs
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class RenderJob:
    """One render of run_render (infer_vipe_all_takes.sh:842-983): a take, one of its ViPE result directories, and the
    dataset roots the paths hang from."""
    take_name: str
    result_subdir: str
    ego_uuid: str
    vipe_results_root: str
    data_dir: str
    meta_files_dir: str
    output_dir: str
    start_frame: int = 0
    end_frame: int = 0
    point_size: float = 5.0
    extra_args: List[str] = field(default_factory=list)

    @property
    def input_dir(self) -> Path:
        return Path(self.vipe_results_root) / self.take_name / self.result_subdir

    @property
    def ego_pose_path(self) -> Path:
        return Path(self.data_dir) / "annotations/ego_pose/test/camera_pose" / f"{self.ego_uuid}.json"

    @property
    def online_calibration_path(self) -> Path:
        return Path(self.data_dir) / "takes" / self.take_name / "trajectory/online_calibration.jsonl"

    @property
    def meta_json_path(self) -> Path:
        return Path(self.meta_files_dir) / f"meta_{self.take_name}_{self.result_subdir}.json"

    @property
    def n_frames(self) -> int:
        return max(self.end_frame - self.start_frame, 0) + 1


def render_argv(job: RenderJob) -> List[str]:
    """Arguments of render_vipe_pointcloud for this job.  The meta.json holds only the frames of the range, so the
    renderer is asked for 0 .. end-start (:956-960)."""
    return ["--input_dir", str(job.input_dir), "--out_dir", str(job.output_dir),
            "--meta_json_path", str(job.meta_json_path), "--point_size", str(job.point_size),
            "--start_frame", "0", "--end_frame", str(max(job.end_frame - job.start_frame, 0)),
            "--fish_eye_rendering", "--use_mean_bg", "--only_bg",
            "--online_calibration_path", str(job.online_calibration_path), *job.extra_args]


def missing_inputs(job: RenderJob) -> List[str]:
    """What run_render checks before it starts the renderer (:871-953), as messages; empty when the job can run."""
    out = []
    if not job.input_dir.is_dir():
        out.append(f"Input result directory not found: {job.input_dir}")
    if not job.ego_pose_path.is_file():
        out.append(f"Ego camera pose not found: {job.ego_pose_path}")
    if not job.online_calibration_path.is_file():
        out.append(f"Online calibration not found: {job.online_calibration_path}")
    return out


def render_one(job: RenderJob) -> int:
    """create_meta_json + the renderer for one job, in this process.  Returns an exit code, never raises."""
    try:
        problems = missing_inputs(job)
        if problems:
            for p in problems:
                logger.error("%s/%s: %s", job.take_name, job.result_subdir, p)
            return EXIT_SKIPPED
        create_meta_json(job.ego_pose_path, job.take_name, job.result_subdir, job.meta_json_path, job.start_frame,
                         job.end_frame)
        from . import render_vipe_pointcloud
        render_vipe_pointcloud.main(render_argv(job))
        return EXIT_OK
    except Exception:  # a failed job must not take the worker (and the jobs queued behind it) down
        logger.error("%s/%s failed:\n%s", job.take_name, job.result_subdir, traceback.format_exc())
        return EXIT_FAILED


def longest_first(jobs: Sequence[RenderJob]) -> List[int]:
    """Job indices by decreasing frame count (ties keep their order): with a shared queue this is the classic
    longest-processing-time rule, which bounds the idle tail of the last GPU by the shortest jobs."""
    return sorted(range(len(jobs)), key=lambda i: (-jobs[i].n_frames, i))


def _worker(gpu: Optional[int], jobs, todo, done, run: Callable[[RenderJob], int]):
    if gpu is not None:
        # before anything in this process touches the GPU: the worker then sees exactly one device, cuda:0
        os.environ["HIP_VISIBLE_DEVICES"] = str(gpu)
    while True:
        i = todo.get()
        if i is None:
            return
        try:
            code = int(run(jobs[i]))
        except Exception:
            logger.error("job %d failed:\n%s", i, traceback.format_exc())
            code = EXIT_FAILED
        done.put((i, gpu, code))


def run_jobs(jobs: Sequence[RenderJob], gpus: Sequence[Optional[int]], run: Callable[[RenderJob], int] = render_one,
             mp_context: str = "spawn") -> List[dict]:
    """Runs every job on one of `gpus` (device indices; None entries make CPU-side workers for tests) and returns, in
    job order, {"job", "gpu", "exit_code"}.  One worker process per entry of `gpus`; workers take the next job when they
    finish one.  Workers are spawned, not forked, so none inherits an initialised GPU runtime."""
    import multiprocessing as mp
    if not gpus:
        raise ValueError("run_jobs needs at least one worker")
    ctx = mp.get_context(mp_context)
    todo, done = ctx.Queue(), ctx.Queue()
    for i in longest_first(jobs):
        todo.put(i)
    for _ in gpus:
        todo.put(None)
    jobs = list(jobs)
    workers = [ctx.Process(target=_worker, args=(g, jobs, todo, done, run), daemon=True) for g in gpus]
    for w in workers:
        w.start()
    results: List[Optional[dict]] = [None] * len(jobs)
    pending = len(jobs)
    while pending:
        try:
            i, gpu, code = done.get(timeout=1.0)
        except Exception:  # queue.Empty: check that somebody is still working
            if not any(w.is_alive() for w in workers) and done.empty():
                break
            continue
        results[i] = {"job": i, "gpu": gpu, "exit_code": code}
        pending -= 1
        state = {EXIT_OK: "done", EXIT_SKIPPED: "skipped"}.get(code, f"failed ({code})")
        logger.info("[%d/%d] %s/%s %s on gpu %s", len(jobs) - pending, len(jobs), jobs[i].take_name,
                    jobs[i].result_subdir, state, gpu)
    for w in workers:
        w.join(timeout=5.0)
    for i, r in enumerate(results):
        if r is None:  # its worker died without reporting
            results[i] = {"job": i, "gpu": None, "exit_code": EXIT_FAILED}
    return results
