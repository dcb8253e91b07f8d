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
  resolution: [%d, %d]
  intrinsics: [%r, %r, %r, %r]
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
