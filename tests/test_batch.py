"""CPU: the dataset-scale driver (egoprior.batch).  Expected values are derived by hand from the rules of
data_preprocess/scripts/infer_vipe_all_takes.sh:696-840 (create_meta_json), :842-983 (run_render) and :1303-1338 (the
process pool); the reference holds no fixtures for them and its shell function was not run, so these are known-answer
tests, not golden vectors (parity unpinned)."""
import json
import os
import time

import pytest

from egoprior import batch


def mat(rows, cols, seed):
    return [[seed + 0.25 * (r * cols + c) for c in range(cols)] for r in range(rows)]


K = [[300.0, 0.0, 224.0], [0.0, 300.0, 224.0], [0.0, 0.0, 1.0]]
CAM01 = {"camera_intrinsics": K, "camera_extrinsics": mat(3, 4, 1.0)}
CAM04 = {"camera_intrinsics": mat(3, 3, 7.0), "camera_extrinsics": mat(3, 4, 2.0)}
FRAMES = {**{str(f): mat(3, 4, 10.0 + f) for f in (0, 1, 2, 3, 5, 8, 13)}, "note": "not a frame"}
ARIA = {"camera_intrinsics": mat(3, 3, 3.0), "camera_extrinsics": FRAMES}
BASE = {"metadata": {"take_uid": "x"}, "cam01": CAM01, "cam04": CAM04}


def test_exo_camera_name():
    assert batch.exo_camera_name("cam04_video_cmu_bike08_2_cam04_2582104") == "cam04"
    assert batch.exo_camera_name("cam1") == "cam1"
    assert batch.exo_camera_name("video_cam02") == "cam03"  # prefix only
    assert batch.exo_camera_name("") == "cam03"


def test_meta_from_frame_dictionary():
    meta = batch.build_meta({**BASE, "aria01": ARIA}, "cam04_video_take_a_cam04_123", 1, 8)
    (s,) = meta["test_datasets"]
    assert s["exo_path"] == "./takes/cam04_video_take_a_cam04_123/cam01.mp4"
    assert s["camera_intrinsics"] == CAM04["camera_intrinsics"] and s["camera_extrinsics"] == CAM04["camera_extrinsics"]
    assert s["ego_intrinsics"] == ARIA["camera_intrinsics"]
    assert s["ego_extrinsics"] == [FRAMES[str(f)] for f in (1, 2, 3, 5, 8)]  # numeric order, inclusive, gaps kept out
    assert list(s.keys()) == ["exo_path", "camera_intrinsics", "camera_extrinsics", "ego_intrinsics", "ego_extrinsics"]
    everything = batch.build_meta({**BASE, "aria01": ARIA}, "cam01_x", 0, 100)["test_datasets"][0]["ego_extrinsics"]
    assert len(everything) == 7
    with pytest.raises(ValueError, match=r"No frames found in range 20-30. Available: 0 ~ 13"):
        batch.build_meta({**BASE, "aria01": ARIA}, "cam01_x", 20, 30)


def test_meta_numeric_key_order_is_not_string_order():
    frames = {str(f): mat(3, 4, float(f)) for f in (10, 9, 100, 2)}
    got = batch.select_ego_extrinsics(frames, 0, 1000)
    assert got == [frames["2"], frames["9"], frames["10"], frames["100"]]


def test_meta_from_lists():
    per_frame = [mat(4, 4, float(i)) for i in range(6)]
    assert batch.select_ego_extrinsics(per_frame, 2, 4) == per_frame[2:5]
    assert batch.select_ego_extrinsics(per_frame, 0, 5) is per_frame
    assert batch.select_ego_extrinsics(per_frame, 0, 99) is per_frame
    assert batch.select_ego_extrinsics(per_frame, 4, 99) == per_frame[4:]
    assert batch.select_ego_extrinsics(per_frame, 7, 9) == []
    one = mat(3, 4, 5.0)  # a bare 3x4: its rows have 4 entries, so it is read as a frame list
    assert batch.select_ego_extrinsics(one, 0, 10) is one
    assert batch.select_ego_extrinsics(one, 1, 1) == [one[1]]
    three = [mat(3, 4, float(i)) for i in range(3)]  # first element has 3 rows: wrapped as one entry
    assert batch.select_ego_extrinsics(three, 0, 2) == [three]
    five = [mat(3, 4, float(i)) for i in range(5)]
    assert batch.select_ego_extrinsics(five, 0, 4) == []
    assert batch.select_ego_extrinsics([], 0, 4) == []
    assert batch.select_ego_extrinsics([1.0, 2.0, 3.0], 0, 4) == [[1.0, 2.0, 3.0]]
    assert batch.select_ego_extrinsics("nonsense", 0, 4) == []


def test_meta_failures_in_reference_order():
    with pytest.raises(ValueError, match=r"Camera 'cam03' not found in ego_pose. Available: \['cam01', 'cam04'\]"):
        batch.build_meta({**BASE, "aria01": ARIA}, "video_without_camera_prefix", 0, 3)
    ok = batch.build_meta({**BASE, "cam03": CAM01, "aria02": ARIA}, "video_without_camera_prefix", 0, 3)
    assert len(ok["test_datasets"][0]["ego_extrinsics"]) == 4
    with pytest.raises(ValueError, match="No aria camera"):
        batch.build_meta(BASE, "cam01_x", 0, 4)
    with pytest.raises(ValueError, match="ego_intrinsics not found"):
        batch.build_meta({**BASE, "aria01": {"camera_extrinsics": FRAMES}}, "cam01_x", 0, 4)
    with pytest.raises(ValueError, match="exo_extrinsics not found for camera cam01"):
        batch.build_meta({"cam01": {"camera_intrinsics": K}, "aria01": ARIA}, "cam01_x", 0, 4)
    with pytest.raises(ValueError, match="exo_intrinsics not found for camera cam01"):
        batch.build_meta({"cam01": {"camera_extrinsics": mat(3, 4, 1.0)}, "aria01": ARIA}, "cam01_x", 0, 4)
    # the empty frame range is reported before a missing exo entry, as the frame selection runs first
    with pytest.raises(ValueError, match="No frames found"):
        batch.build_meta({"cam01": {}, "aria01": ARIA}, "cam01_x", 50, 60)
    with pytest.raises(ValueError, match="No ego_extrinsics found for frames 0-4"):
        batch.build_meta({**BASE, "aria01": {"camera_intrinsics": K, "camera_extrinsics": []}}, "cam01_x", 0, 4)
    with pytest.raises(ValueError, match="No ego_extrinsics found"):
        batch.build_meta({**BASE, "aria01": {"camera_intrinsics": K}}, "cam01_x", 0, 4)
    # the first key that starts with "aria" in file order is the ego camera
    first = batch.build_meta({**BASE, "aria07": {"camera_intrinsics": K, "camera_extrinsics": FRAMES},
                              "aria01": ARIA}, "cam01_x", 0, 0)
    assert first["test_datasets"][0]["ego_intrinsics"] == K


def test_create_meta_json_file_and_loader_round_trip(tmp_path):
    pose = tmp_path / "pose" / "uuid.json"
    pose.parent.mkdir()
    pose.write_text(json.dumps({**BASE, "aria01": ARIA}))
    out = tmp_path / "meta" / "deep" / "meta_take_a_cam04_x.json"
    meta = batch.create_meta_json(pose, "take_a", "cam04_x", out, 2, 5)
    assert json.loads(out.read_text()) == meta
    assert out.read_text().startswith('{\n  "test_datasets": [\n    {\n      "exo_path"')  # indent 2
    # the renderer finds the sample by the stem of its input directory
    from egoprior.render_vipe_pointcloud import load_camera_params_from_meta
    K_exo, T_exo, K_ego, ego = load_camera_params_from_meta(str(out), str(tmp_path / "vipe" / "take_a" / "cam04_x"))
    assert K_exo.shape == (3, 3) and T_exo.shape == (4, 4) and K_ego.shape == (3, 3) and len(ego) == 3
    assert T_exo[3].tolist() == [0.0, 0.0, 0.0, 1.0] and T_exo[0, 0] == 2.0
    with pytest.raises(FileNotFoundError, match="Ego camera pose not found"):
        batch.create_meta_json(tmp_path / "nope.json", "take_a", "cam04_x", out, 0, 1)


def job(tmp_path, take, sub, start=0, end=48, **kw):
    return batch.RenderJob(take_name=take, result_subdir=sub, ego_uuid="u-" + take,
                           vipe_results_root=str(tmp_path / "vipe"), data_dir=str(tmp_path / "data"),
                           meta_files_dir=str(tmp_path / "meta"), output_dir=str(tmp_path / "out"), start_frame=start,
                           end_frame=end, **kw)


def test_render_argv_and_paths(tmp_path):
    j = job(tmp_path, "cmu_bike08_2", "cam04_video_x", start=10, end=58, point_size=3.0)
    assert j.input_dir == tmp_path / "vipe/cmu_bike08_2/cam04_video_x"
    assert j.ego_pose_path == tmp_path / "data/annotations/ego_pose/test/camera_pose/u-cmu_bike08_2.json"
    assert j.online_calibration_path == tmp_path / "data/takes/cmu_bike08_2/trajectory/online_calibration.jsonl"
    assert j.meta_json_path == tmp_path / "meta/meta_cmu_bike08_2_cam04_video_x.json"
    argv = batch.render_argv(j)
    from egoprior.render_vipe_pointcloud import get_parser
    a = get_parser().parse_args(argv)
    assert (a.start_frame, a.end_frame, a.point_size) == (0, 48, 3.0)
    assert a.fish_eye_rendering and a.use_mean_bg and a.only_bg
    assert a.online_calibration_path == str(j.online_calibration_path) and a.input_dir == str(j.input_dir)
    assert batch.render_argv(job(tmp_path, "t", "cam01_x", start=5, end=2))[9] == "0"  # negative span clamps to 0
    assert len(batch.missing_inputs(j)) == 3
    assert batch.render_one(j) == batch.EXIT_SKIPPED


def test_longest_first_order(tmp_path):
    jobs = [job(tmp_path, "a", "cam01_x", 0, 9), job(tmp_path, "b", "cam01_x", 0, 99), job(tmp_path, "c", "cam01_x", 5, 14),
            job(tmp_path, "d", "cam01_x", 0, 49)]
    assert [j.n_frames for j in jobs] == [10, 100, 10, 50]
    assert batch.longest_first(jobs) == [1, 3, 0, 2]


def fake_render(j):
    """Stands in for the renderer in a worker: records which worker ran the job and returns a code by take name."""
    time.sleep(0.01 * j.n_frames)
    with open(os.path.join(j.output_dir, f"{j.take_name}.txt"), "w") as f:
        f.write(f"{os.getpid()} {os.environ.get('HIP_VISIBLE_DEVICES', '-')}")
    if j.take_name.startswith("boom"):
        raise RuntimeError("worker-side failure")
    return batch.EXIT_SKIPPED if j.take_name.startswith("skip") else batch.EXIT_OK


def test_run_jobs_two_workers(tmp_path):
    (tmp_path / "out").mkdir()
    names = ["t0", "t1", "skip2", "t3", "boom4", "t5", "t6"]
    jobs = [job(tmp_path, n, "cam01_x", 0, 3 + 2 * i) for i, n in enumerate(names)]
    res = batch.run_jobs(jobs, gpus=[None, None], run=fake_render)
    assert [r["job"] for r in res] == list(range(len(jobs)))
    assert [r["exit_code"] for r in res] == [0, 0, 2, 0, 1, 0, 0]
    pids = {(tmp_path / "out" / f"{n}.txt").read_text().split()[0] for n in names}
    assert len(pids) == 2  # both workers took part, and only they
    with pytest.raises(ValueError):
        batch.run_jobs(jobs, gpus=[])


def test_run_jobs_pins_one_device_per_worker(tmp_path):
    (tmp_path / "out").mkdir()
    jobs = [job(tmp_path, f"t{i}", "cam01_x", 0, 2) for i in range(4)]
    res = batch.run_jobs(jobs, gpus=[5, 6], run=fake_render)
    assert all(r["exit_code"] == 0 for r in res)
    for r, j in zip(res, jobs):
        pid, visible = (tmp_path / "out" / f"{j.take_name}.txt").read_text().split()
        assert visible == str(r["gpu"]) and r["gpu"] in (5, 6)
