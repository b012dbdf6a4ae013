"""GPU: the dataset-scale driver end to end -- several (take, exo camera) jobs on a synthetic Ego-Exo4D-shaped tree go
through worker processes and must produce the same bits as the same command lines run in this process (which
tests/test_gpu_dropin.py checks against the CPU oracle)."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32


def test_jobs_over_worker_processes(tmp_path):
    from egoprior import batch, io as eio, render_vipe_pointcloud as rv, synth
    fx = synth.load_camera_fixture()
    exo = synth.exo_w2c(fx)
    F, H, W = 5, 72, 128
    jobs = []
    for t, take in enumerate(("take_a", "take_b")):
        egos = synth.ego_orbit_w2c(F + 3, exo)
        pose = {"metadata": {"take": take},
                "cam01": {"camera_intrinsics": fx["cam01"]["camera_intrinsics"], "camera_extrinsics": exo[:3].tolist()},
                "cam04": {"camera_intrinsics": fx["cam01"]["camera_intrinsics"], "camera_extrinsics": exo[:3].tolist()},
                "aria01": {"camera_intrinsics": [[150.0, 0, 224.0], [0, 150.0, 224.0], [0, 0, 1]],
                           "camera_extrinsics": {str(i): e[:3].tolist() for i, e in enumerate(egos)}}}
        pose_path = tmp_path / "data/annotations/ego_pose/test/camera_pose" / f"uuid-{take}.json"
        pose_path.parent.mkdir(parents=True, exist_ok=True)
        pose_path.write_text(json.dumps(pose))
        calib = tmp_path / "data/takes" / take / "trajectory/online_calibration.jsonl"
        calib.parent.mkdir(parents=True)
        calib.write_text(json.dumps(fx["online_calibration_line0"]) + "\n")
        for cam in ("cam01", "cam04"):
            sub = f"{cam}_video_{take}_{cam}_1"
            if (take, cam) != ("take_b", "cam04"):  # that one has no ViPE result: its job is skipped, not failed
                # F - t frames: the renderer insists that the requested range is the clip's range
                d, rgb, m = [x.numpy() for x in synth.make_frames(H, W, range(t, F), 12)]
                # take_b stores its depth the way ViPE does, as a zip of EXR files, take_a as the array twin
                eio.write_synthetic_artifacts(tmp_path / "vipe" / take / sub, "exo", d, rgb, m,
                                              synth.exo_intrinsics(H, W), np.eye(4, dtype=F32),
                                              depth_as_exr_zip=(take == "take_b"))
            jobs.append(batch.RenderJob(take_name=take, result_subdir=sub, ego_uuid=f"uuid-{take}",
                                        vipe_results_root=str(tmp_path / "vipe"), data_dir=str(tmp_path / "data"),
                                        meta_files_dir=str(tmp_path / "meta"), output_dir=str(tmp_path / "out"),
                                        start_frame=2, end_frame=2 + F - 1 - t, point_size=4.0))
    res = batch.run_jobs(jobs, gpus=[0, 0])
    assert [r["exit_code"] for r in res] == [0, 0, 0, batch.EXIT_SKIPPED]
    for j in jobs[:3]:
        meta = json.loads(j.meta_json_path.read_text())["test_datasets"][0]
        assert len(meta["ego_extrinsics"]) == j.n_frames
        got = tmp_path / "out" / j.result_subdir
        if (got / "ego_Prior.mp4").exists():
            continue  # lossy container: nothing to compare bit for bit
        frames = np.load(got / "ego_Prior.npy")
        assert frames.shape == (j.n_frames, 448, 448, 3) and frames.any()
        argv = batch.render_argv(j)
        argv[argv.index("--out_dir") + 1] = str(tmp_path / "again")
        rv.main(argv)
        assert np.array_equal(frames, np.load(tmp_path / "again" / j.result_subdir / "ego_Prior.npy"))
