"""EuRoC harness (rd_vio_amd/euroc.py): mav0 reader / writer round trip, the cv::undistort-equivalent
remap, YAML -> pipeline config, TUM writer, ATE.  There is no EuRoC data in this image: the stream is synthetic, written
in the EuRoC layout."""
import ctypes
import os

import numpy as np
import pytest

import pipeline_util as pu
from rd_vio_amd import euroc, synth

W, H = 376, 240
K = synth.EUROC_K.copy()
K[:2] *= 0.5

SENSOR_YAML = """%%YAML:1.0
imu:
  extrinsic:
    q_bi: [ 0.0, 0.0, 0.0, 1.0 ]
    p_bi: [ 0.0, 0.0, 0.0 ]
  noise:
    cov_g: [2.8791302399999997e-08, 0.0, 0.0, 0.0, 2.8791302399999997e-08, 0.0, 0.0, 0.0, 2.8791302399999997e-08]
    cov_a: [4.0e-6, 0.0, 0.0, 0.0, 4.0e-6, 0.0, 0.0, 0.0, 4.0e-6]
    cov_bg: [3.7608844899999997e-10, 0.0, 0.0, 0.0, 3.7608844899999997e-10, 0.0, 0.0, 0.0, 3.7608844899999997e-10]
    cov_ba: [9.0e-6, 0.0, 0.0, 0.0, 9.0e-6, 0.0, 0.0, 0.0, 9.0e-6]
cam0:
  resolution: [%d, %d]        # resolution of camera
  camera_model: pinhole
  intrinsics: [%r, %r, %r, %r] # fu, fv, cu, cv
  camera_distortion_flag: 0
  distortion: [0.0, 0.0, 0.0, 0.0] # k1, k2, p1, p2
  time_offset: 0.0
  extrinsic:
    q_bc: [ -7.7071797555374275e-03, 1.0499323370587278e-02, 7.0175280029197162e-01, 7.1230146066895372e-01 ]
    p_bc: [ -0.0216401454975, -0.064676986768, 0.00981073058949 ]
  noise: [0.5, 0.0, 0.0, 0.5]
""" % (W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))

SETTING_YAML = """%YAML:1.0
sliding_window:
  size: 8
  subframe_size: 3
  force_keyframe_landmarks: 50
feature_tracker:
  min_keypoint_distance: 10.0
  max_keypoint_detection: 150
  max_frames: 20
  predict_keypoints: true
initializer:
  keyframe_num: 8
  keyframe_gap: 2
solver:
  iteration_limit: 30
rotation:
  misalignment_threshold: 0.02
  ransac_threshold: 10
parsac:
  parsac_flag: false
"""


@pytest.fixture(scope="module")
def mav(tmp_path_factory):
    d = tmp_path_factory.mktemp("mav0")
    frames, ts, imu, gt = synth.make_stream(32, W, H, K)
    euroc.write_mav0(str(d), frames, ts, imu, gt, K)
    (d / "sensor.yaml").write_text(SENSOR_YAML)
    (d / "setting.yaml").write_text(SETTING_YAML)
    return d, frames, ts, imu, gt


def test_mav0_round_trip(mav):
    d, frames, ts, imu, gt = mav
    ds = euroc.EurocDataset(str(d))
    imgs = [c for c in ds.clips if "image" in c]
    assert len(imgs) == len(ts) and np.allclose([c["t"] for c in imgs], ts, atol=1e-9)
    assert (ds.read_image(imgs[3]) == frames[3]).all()                       # zero distortion: images come back untouched
    rows = np.array([[c["t"], *c["gyro"], *c["acc"]] for c in ds.clips if "gyro" in c])
    assert np.allclose(rows, imu, atol=1e-9)
    assert [c["ns"] for c in ds.clips] == sorted(c["ns"] for c in ds.clips)  # ordered by time
    g = ds.init_states_at(ts[:5])
    assert np.allclose(g, gt[:5], atol=1e-9)


def test_config_from_yaml(mav):
    d, *_ = mav
    Kc, w, h, extr, noise, over = euroc.config_overrides(str(d / "sensor.yaml"), str(d / "setting.yaml"))
    assert (w, h) == (W, H) and np.allclose(Kc, K)
    assert np.allclose(extr, synth.EUROC_EXTR) and np.allclose(noise, synth.EUROC_NOISE)
    assert over["sliding_window_size"] == 8 and over["initializer_keyframe_gap"] == 2 and over["parsac_flag"] == 0
    assert over["feature_tracker_predict_keypoints"] == 1 and over["keypoint_noise_cov"] == [0.5, 0.0, 0.0, 0.5]


def test_undistort_matches_the_radtan_model():
    Kf = synth.EUROC_K
    dist = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]
    mx, my = euroc.undistort_map(Kf, dist, 752, 480)
    # a pixel at the principal point is a fixed point; elsewhere the map equals the forward distortion model
    cx, cy = int(round(Kf[0, 2])), int(round(Kf[1, 2]))
    assert abs(mx[cy, cx] - cx) < 0.05 and abs(my[cy, cx] - cy) < 0.05
    u, v = 100, 60
    x, y = (u - Kf[0, 2]) / Kf[0, 0], (v - Kf[1, 2]) / Kf[1, 1]
    r2 = x * x + y * y
    rad = 1 + dist[0] * r2 + dist[1] * r2 * r2
    xd = x * rad + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
    yd = y * rad + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y
    assert abs(mx[v, u] - (Kf[0, 0] * xd + Kf[0, 2])) < 1e-9 and abs(my[v, u] - (Kf[1, 1] * yd + Kf[1, 2])) < 1e-9
    # identity map reproduces the image; a half-pixel shift interpolates
    img = (np.arange(480)[:, None] * 3 + np.arange(752)[None, :]).astype(np.float64) % 256
    img = img.astype(np.uint8)
    vv, uu = np.mgrid[0:480, 0:752].astype(np.float64)
    assert (euroc.remap_bilinear(img, uu, vv) == img).all()
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8), (10, 1))
    vv, uu = np.mgrid[0:10, 0:99].astype(np.float64)
    assert (euroc.remap_bilinear(ramp, uu + 0.5, vv)[:, :98] == ramp[:, :98] + 1).all()


def test_replay_tum_and_ate(mav, tmp_path):
    d, frames, ts, imu, gt = mav
    lib, shim = pu.load_pipeline_lib(), pu.build_oracle_backend()
    Kc, w, h, extr, noise, over = euroc.config_overrides(str(d / "sensor.yaml"), str(d / "setting.yaml"))
    cfg = euroc.apply_overrides(pu.default_config(lib, Kc, w, h, extr, noise), over)
    hnd = ctypes.c_void_p()
    assert pu.oracle_pipeline_factory(lib, shim, cfg)(ctypes.byref(hnd)) == 0
    ds = euroc.EurocDataset(str(d))
    init = np.ascontiguousarray(ds.init_states_at(ts))
    lib.rdvio_pipeline_set_init_states(hnd, len(init), init.ctypes.data_as(ctypes.c_void_p))
    traj, spent = euroc.replay(lib, hnd, ds)
    lib.rdvio_pipeline_destroy(hnd)
    assert len(traj) >= 10 and spent > 0
    out = tmp_path / "traj.txt"
    euroc.write_tum(str(out), traj)
    back = np.loadtxt(str(out))
    assert back.shape == traj.shape and np.allclose(back, traj, atol=1e-8)
    p_gt = ds.init_states_at(traj[:, 0])[:, 5:8]
    assert euroc.ate_rmse(traj[:, 1:4], p_gt) < 0.05
    # ATE is invariant to a rigid motion of the estimate
    Rz = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    assert abs(euroc.ate_rmse(traj[:, 1:4] @ Rz.T + [1, 2, 3], p_gt) - euroc.ate_rmse(traj[:, 1:4], p_gt)) < 1e-9


# ------------------------------------------------------------------------------------------------ C++ boundary
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _yaml_tool():
    import subprocess

    exe = os.path.join(ROOT, "tests", "cpp", "yaml_config_test.bin")
    src = os.path.join(ROOT, "tests", "cpp", "yaml_config_test.cpp")
    libdir = os.path.join(ROOT, "rd_vio_amd")
    from rd_vio_amd import build as rbuild

    rbuild.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, src, "-L", libdir, "-lrdvio_pipeline", "-lrdvio_hip", f"-Wl,-rpath,{libdir}",
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return lambda config, calib: subprocess.run([exe, str(config), str(calib)], capture_output=True, text=True, timeout=60).stdout.strip()


def test_cpp_yaml_loader_matches_the_python_mapping(mav):
    """rdvio_hip::load_yaml_config (rd_vio_amd/host/rdvio_yaml.hpp; yaml_config.cpp:83-338 without yaml-cpp) against the Python
    mapping of the same two files, field by field, incl. the defaults of rdvio::Config for keys the files omit."""
    import json

    d, *_ = mav
    tool = _yaml_tool()
    got = json.loads(tool(d / "setting.yaml", d / "sensor.yaml"))
    lib = pu.load_pipeline_lib()
    Kc, w, h, extr, noise, over = euroc.config_overrides(str(d / "sensor.yaml"), str(d / "setting.yaml"))
    cfg = euroc.apply_overrides(pu.default_config(lib, Kc, w, h, extr, noise), over)
    for name, v in got.items():
        if name == "extras":
            continue
        want = getattr(cfg, name)
        if isinstance(v, list):
            assert list(want) == v, name
        else:
            assert want == v, (name, want, v)
    assert got["sliding_window_size"] == 8 and got["feature_tracker_max_keypoint_detection"] == 150 and got["parsac_flag"] == 0
    assert got["parsac_keyframe_check_size"] == 3 and got["feature_tracker_clahe_width"] == 8      # rdvio::Config defaults
    assert got["extras"]["solver_time_limit"] == 1.0e6 and got["extras"]["camera_distortion_flag"] == 0


def test_cpp_yaml_loader_reference_layout_and_errors(tmp_path):
    """files laid out like configs/euroc_sensor.yaml / configs/setting.yaml (multi-line flow sequences, comments, unused keys,
    a nested T_BS block, `true` / scientific notation) and the four exception kinds with the reference's messages
    (yaml_config.h:10-27)."""
    import json

    tool = _yaml_tool()
    sensor = tmp_path / "sensor.yaml"
    setting = tmp_path / "setting.yaml"
    sensor.write_text("""%YAML:1.0
imu:
  # inertial sensor noise model parameters (static)
  gyroscope_noise_density: 0.01       # [ rad / s / sqrt(Hz) ]
  accelerometer_bias: [0.0, 0.0, 0.0] # acc bias prior
  extrinsic:
    q_bi: [ 0.0, 0.0, 0.0, 1.0 ] # x y z w
    p_bi: [ 0.0, 0.0, 0.0 ] # x y z [m]
  noise:
    cov_g: [
      2.5e-08, 0.0, 0.0,
      0.0, 2.5e-08, 0.0,
      0.0, 0.0, 2.5e-08]
    cov_a: [
      4.0e-6, 0.0, 0.0,
      0.0, 4.0e-6, 0.0,
      0.0, 0.0, 4.0e-6]
    cov_bg: [1e-10, 0, 0, 0, 1e-10, 0, 0, 0, 1e-10]
    cov_ba: [9.0e-6, 0.0, 0.0, 0.0, 9.0e-6, 0.0, 0.0, 0.0, 9.0e-6]

cam0:
  # camera0 wrt. body frame
  T_BS:
    cols: 4
    rows: 4
    data: [1.0, 0.0, 0.0, 0.5,
           0.0, 1.0, 0.0, 0.25,
           0.0, 0.0, 1.0, 0.125,
           0.0, 0.0, 0.0, 1.0]
  resolution: [752, 480]        # resolution of camera
  camera_model: pinhole         # camera model
  intrinsics: [458.654, 457.296, 367.215, 248.375] # fu, fv, cu, cv
  camera_distortion_flag: 1     # use distortion model or not
  distortion: [-0.28, 0.07, 0.0002, 1.76e-05] # k1, k2, p1, p2, xi
  time_offset: 0.0              # camera time delay wrt. IMU
  extrinsic:
    q_bc: [ 0.0, 0.0, 0.7071067811865476, 0.7071067811865476 ] # x y z w
    p_bc: [ -0.02, -0.06, 0.01 ] # x y z [m]
  noise: [
    0.5, 0.0,
    0.0, 0.5] # [pixel^2]
""")
    setting.write_text("""%YAML:1.0
output:
  q_bo: [ 0.0, 0.0, 0.0, 1.0 ] # x y z w
  p_bo: [ 0.0, 0.0, 0.0 ] # x y z [m]

sliding_window:
  size: 12 # 10 by default
  subframe_size: 5 # 3 by default

feature_tracker:
  max_keypoint_detection: 200
  predict_keypoints: true
  clahe_clip_limit: 6.0

solver:
  iteration_limit: 30
  time_limit: 1.0e6 # [s]

parsac:
  parsac_flag: true
  dynamic_probability: 0.15
  keyframe_check_size: 1
""")
    got = json.loads(tool(setting, sensor))
    assert (got["width"], got["height"]) == (752, 480) and got["K"] == [458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1]
    assert got["gyroscope_noise_cov"][0] == 2.5e-08 and got["gyroscope_noise_cov"][8] == 2.5e-08 and got["gyroscope_bias_noise_cov"][4] == 1e-10
    assert got["q_bc"][2] == 0.7071067811865476 and got["p_bc"] == [-0.02, -0.06, 0.01] and got["keypoint_noise_cov"] == [0.5, 0, 0, 0.5]
    assert got["sliding_window_size"] == 12 and got["sliding_window_subframe_size"] == 5 and got["solver_iteration_limit"] == 30
    assert got["parsac_flag"] == 1 and got["parsac_keyframe_check_size"] == 1 and got["extras"]["parsac_dynamic_probability"] == 0.15
    assert got["extras"]["camera_distortion"][0] == -0.28 and got["extras"]["camera_distortion_flag"] == 1
    assert got["sliding_window_force_keyframe_landmarks"] == 35 and got["initializer_keyframe_num"] == 8     # config.cpp defaults
    # the four exception kinds
    assert tool(setting, tmp_path / "nope.yaml") == f"EXCEPTION load: cannot load config {tmp_path / 'nope.yaml'}"
    broken = tmp_path / "missing.yaml"
    broken.write_text(sensor.read_text().replace("  intrinsics: [458.654, 457.296, 367.215, 248.375] # fu, fv, cu, cv\n", ""))
    assert tool(setting, broken) == 'EXCEPTION missing: config "cam0.intrinsics" is mandatory'
    broken.write_text(sensor.read_text().replace("[752, 480]", "[752, 480, 3]"))
    assert tool(setting, broken) == 'EXCEPTION type: config "cam0.resolution" has wrong type'
    bad_setting = tmp_path / "bad_setting.yaml"
    bad_setting.write_text(setting.read_text().replace("size: 12", "size: [12]"))
    assert tool(bad_setting, sensor) == 'EXCEPTION type: config "sliding_window.size" has wrong type'
    bad_setting.write_text("sliding_window:\n    size: 3\n  subframe_size: 2\n")
    assert tool(bad_setting, sensor).startswith("EXCEPTION parse:")


@pytest.mark.gpu
def test_cpp_test_euroc_matches_run_euroc(mav, tmp_path):
    """rd_vio_amd/test_euroc (C++: rdvio_hip::Odometry(calib, config) + the mav0 reader of host/test_euroc.cpp, i.e. the
    reference's examples/test_euroc.cpp loop) and scripts/run_euroc.py (Python harness) over the same synthetic mav0 tree:
    the same poses at the same times -- equal to the files' 9-decimal precision (one unit in the last place at most: the two
    harnesses hand the pipeline bootstrap states that may differ by an ulp)."""
    import json
    import subprocess
    import sys

    d, frames, ts, imu, gt = mav
    exe = os.path.join(ROOT, "rd_vio_amd", "test_euroc")
    assert os.path.exists(exe)
    a = subprocess.run([exe, str(d), str(d / "sensor.yaml"), str(d / "setting.yaml"), "--out", str(tmp_path / "cpp.txt"), "--bootstrap-from-groundtruth"],
                       capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    rep = json.loads(a.stdout.strip().splitlines()[-1])
    b = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_euroc.py"), str(d), "--sensor", str(d / "sensor.yaml"), "--setting",
                        str(d / "setting.yaml"), "--out", str(tmp_path / "py.txt"), "--bootstrap-from-groundtruth"], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stdout + b.stderr
    rep_py = json.loads(b.stdout.strip().splitlines()[-1])
    assert rep["frames"] == len(ts) and rep["poses"] == rep_py["poses"] >= 10 and rep["state"] == 1
    # Odometry::local_map: landmarks of the window with the reference's axis swap applied by the mirror class (rdvio.hpp:91-97)
    assert rep["local_map_points"] >= 20 and rep["local_map_axis_swap_ok"] and rep["transform_world_cam_last_row_ok"]
    ta, tb = np.loadtxt(str(tmp_path / "cpp.txt")), np.loadtxt(str(tmp_path / "py.txt"))
    assert ta.shape == tb.shape and (ta[:, 0] == tb[:, 0]).all()
    assert np.abs(ta - tb).max() <= 2.5e-9, np.abs(ta - tb).max()
    assert rep_py["ate_rmse_m"] < 0.05
    # the constructor's failure modes surface as the reference's exception messages
    c = subprocess.run([exe, str(d), str(d / "nope.yaml"), str(d / "setting.yaml")], capture_output=True, text=True, timeout=60)
    assert c.returncode == 1 and "cannot load config" in c.stderr


def test_yaml_settings_the_pipeline_cannot_ignore(mav, tmp_path):
    """initializer.refine_imu reaches the pipeline configuration (initializer.cpp:373 skips the gravity refinement when false), and a
    solver.time_limit that could cut a solve short is refused loudly: the device solver is bounded by the iteration limit only."""
    import json

    d, *_ = mav
    tool = _yaml_tool()
    base = (d / "setting.yaml").read_text()
    s1 = tmp_path / "no_refine.yaml"
    s1.write_text(base.replace("initializer:\n", "initializer:\n  refine_imu: false\n"))
    got = json.loads(tool(s1, d / "sensor.yaml"))
    assert got["initializer_refine_imu"] == 0 and got["extras"]["initializer_refine_imu"] == 0
    assert json.loads(tool(d / "setting.yaml", d / "sensor.yaml"))["initializer_refine_imu"] == 1
    s2 = tmp_path / "time_limit.yaml"
    s2.write_text(base.replace("solver:\n", "solver:\n  time_limit: 0.01\n"))
    out = tool(s2, d / "sensor.yaml")
    assert out.startswith("EXCEPTION parse:") and "solver.time_limit" in out
