"""EuRoC `mav0` harness: dataset reader, cv::undistort-equivalent remap, YAML configs -> pipeline
config, the test_euroc replay loop, TUM trajectory writer and ATE.

Reference behaviour followed (file:line under /root/reference):
  examples/dataset.hpp:454-624   EuRoC reader: cam0/data.csv (timestamp [ns], filename), imu0/data.csv (timestamp, gyro xyz,
                                 acc xyz), cam0/sensor.yaml (resolution, intrinsics, distortion_coefficients), images
                                 undistorted on load (:585-592), clips ordered by timestamp
  examples/test_euroc.cpp:46-95  replay loop: addMotion for clips with IMU data, then addFrame for clips with an image
  configs/euroc_sensor.yaml, configs/setting.yaml via src/rdvio_extra/src/yaml_config.cpp: keys -> rdvio::Config
The reference's trajectory output goes to a viewer; the TUM text format (t tx ty tz qx qy qz qw) and the Umeyama-aligned
ATE are this build's (SURVEY.md 8d metric 2).  PNG decoding uses PIL; there is no EuRoC data in this image, so the tests
replay a synthetic stream written in the same layout (write_mav0).
"""
import csv
import os

import numpy as np


# ------------------------------------------------------------------------------------------------ configuration
def _load_yaml(path):
    import yaml

    text = open(path).read()
    if text.startswith("%YAML"):
        text = text.split("\n", 1)[1]          # OpenCV-style directive line
    return yaml.safe_load(text)


def config_overrides(sensor_yaml, setting_yaml=None):
    """(K, width, height, extr14, noise36, overrides) from the reference's two YAML files (yaml_config.cpp)."""
    s = _load_yaml(sensor_yaml)
    cam, imu = s["cam0"], s["imu"]
    fu, fv, cu, cv_ = [float(v) for v in cam["intrinsics"]]
    K = np.array([[fu, 0, cu], [0, fv, cv_], [0, 0, 1.0]])
    w, h = cam["resolution"]
    fl = lambda seq: [float(v) for v in seq]  # noqa: E731  (YAML 1.1 reads "1e-5" as a string)
    extr = np.array(fl(cam["extrinsic"]["q_bc"]) + fl(cam["extrinsic"]["p_bc"]) + fl(imu["extrinsic"]["q_bi"]) + fl(imu["extrinsic"]["p_bi"]))
    nz = imu["noise"]
    noise = np.array(fl(nz["cov_g"]) + fl(nz["cov_a"]) + fl(nz["cov_bg"]) + fl(nz["cov_ba"]))
    over = {"keypoint_noise_cov": [float(v) for v in cam["noise"]]}
    if setting_yaml:
        t = _load_yaml(setting_yaml)
        names = {
            ("sliding_window", "size"): "sliding_window_size", ("sliding_window", "subframe_size"): "sliding_window_subframe_size",
            ("sliding_window", "force_keyframe_landmarks"): "sliding_window_force_keyframe_landmarks",
            ("feature_tracker", "min_keypoint_distance"): "feature_tracker_min_keypoint_distance",
            ("feature_tracker", "max_keypoint_detection"): "feature_tracker_max_keypoint_detection",
            ("feature_tracker", "max_init_frames"): "feature_tracker_max_init_frames",
            ("feature_tracker", "max_frames"): "feature_tracker_max_frames",
            ("feature_tracker", "predict_keypoints"): "feature_tracker_predict_keypoints",
            ("feature_tracker", "clahe_clip_limit"): "feature_tracker_clahe_clip_limit",
            ("feature_tracker", "clahe_width"): "feature_tracker_clahe_width", ("feature_tracker", "clahe_height"): "feature_tracker_clahe_height",
            ("initializer", "keyframe_num"): "initializer_keyframe_num", ("initializer", "keyframe_gap"): "initializer_keyframe_gap",
            ("initializer", "min_matches"): "initializer_min_matches", ("initializer", "min_parallax"): "initializer_min_parallax",
            ("initializer", "min_triangulation"): "initializer_min_triangulation", ("initializer", "min_landmarks"): "initializer_min_landmarks",
            ("initializer", "refine_imu"): "initializer_refine_imu",
            ("solver", "iteration_limit"): "solver_iteration_limit",
            ("rotation", "misalignment_threshold"): "rotation_misalignment_threshold", ("rotation", "ransac_threshold"): "rotation_ransac_threshold",
            ("parsac", "parsac_flag"): "parsac_flag", ("parsac", "keyframe_check_size"): "parsac_keyframe_check_size",
        }
        for (sec, key), field in names.items():
            if sec in t and key in t[sec]:
                v = t[sec][key]
                over[field] = int(v) if isinstance(v, bool) else (float(v) if isinstance(v, str) else v)
        if "output" in t:
            over["q_bo"] = [float(v) for v in t["output"].get("q_bo", [0, 0, 0, 1])]
            over["p_bo"] = [float(v) for v in t["output"].get("p_bo", [0, 0, 0])]
    return K, int(w), int(h), extr, noise, over


def apply_overrides(cfg, over):
    for k, v in over.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            getattr(cfg, k)[:] = list(v)
        else:
            setattr(cfg, k, v)
    return cfg


# ------------------------------------------------------------------------------------------------ undistortion
def undistort_map(K, dist, w, h, new_K=None):
    """source coordinates (map_x, map_y) of cv::undistort(img, K, dist): for every destination pixel, normalise with
    the new camera matrix (= K by default), apply the radial-tangential model (k1, k2, p1, p2[, k3]), re-project with K."""
    new_K = K if new_K is None else new_K
    k1, k2, p1, p2 = [float(v) for v in dist[:4]]
    k3 = float(dist[4]) if len(dist) > 4 else 0.0
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    x = (u - new_K[0, 2]) / new_K[0, 0]
    y = (v - new_K[1, 2]) / new_K[1, 1]
    r2 = x * x + y * y
    radial = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]


def remap_bilinear(img, map_x, map_y):
    """cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) in floating point (OpenCV interpolates with 5-bit fixed-point weights:
    results can differ by one grey level; unpinned)."""
    h, w = img.shape
    x0 = np.floor(map_x).astype(np.int64)
    y0 = np.floor(map_y).astype(np.int64)
    fx = map_x - x0
    fy = map_y - y0

    def at(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        out = np.zeros(map_x.shape, dtype=np.float64)
        out[ok] = img[yy[ok], xx[ok]]
        return out

    val = (at(y0, x0) * (1 - fx) + at(y0, x0 + 1) * fx) * (1 - fy) + (at(y0 + 1, x0) * (1 - fx) + at(y0 + 1, x0 + 1) * fx) * fy
    return np.clip(np.rint(val), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ dataset
class EurocDataset:
    """mav0 directory -> time-ordered clips {t [s], gyro, acc, image path}; images are read and undistorted on access."""

    def __init__(self, mav_dir):
        self.dir = mav_dir
        cam_yaml = _load_yaml(os.path.join(mav_dir, "cam0", "sensor.yaml"))
        fu, fv, cu, cv_ = [float(v) for v in cam_yaml["intrinsics"]]
        self.K = np.array([[fu, 0, cu], [0, fv, cv_], [0, 0, 1.0]])
        self.dist = [float(v) for v in cam_yaml.get("distortion_coefficients", [0, 0, 0, 0])]
        self.width, self.height = [int(v) for v in cam_yaml["resolution"]]
        self._map = None if not any(self.dist) else undistort_map(self.K, self.dist, self.width, self.height)
        clips = {}
        with open(os.path.join(mav_dir, "cam0", "data.csv")) as f:
            for row in csv.reader(f):
                if not row or row[0].startswith("#"):
                    continue
                clips.setdefault(int(row[0]), {})["image"] = os.path.join(mav_dir, "cam0", "data", f"{int(row[0])}.png")
        with open(os.path.join(mav_dir, "imu0", "data.csv")) as f:
            for row in csv.reader(f):
                if not row or row[0].startswith("#"):
                    continue
                v = [float(x) for x in row[1:7]]
                c = clips.setdefault(int(row[0]), {})
                c["gyro"], c["acc"] = np.array(v[0:3]), np.array(v[3:6])
        self.clips = [dict(ns=ns, t=ns / 1e9, **clips[ns]) for ns in sorted(clips)]
        self.groundtruth = None
        gt_csv = os.path.join(mav_dir, "state_groundtruth_estimate0", "data.csv")
        if os.path.exists(gt_csv):
            rows = [[float(x) for x in r] for r in csv.reader(open(gt_csv)) if r and not r[0].startswith("#")]
            g = np.array(rows)  # t[ns], p(3), q(w,x,y,z), v(3), bw(3), ba(3)
            self.groundtruth = np.column_stack([g[:, 0] / 1e9, g[:, 5:8], g[:, 4], g[:, 1:4], g[:, 8:11], g[:, 11:14], g[:, 14:17]])

    def read_image(self, clip):
        from PIL import Image

        img = np.asarray(Image.open(clip["image"]).convert("L"))
        return img if self._map is None else remap_bilinear(img, *self._map)

    def init_states_at(self, times):
        """ground-truth rows (t, q, p, v, bg, ba) interpolated at the given times (bootstrap states)."""
        g = self.groundtruth
        out = np.zeros((len(times), 17))
        for i, t in enumerate(times):
            j = int(np.clip(np.searchsorted(g[:, 0], t), 1, len(g) - 1))
            a = (t - g[j - 1, 0]) / (g[j, 0] - g[j - 1, 0])
            row = (1 - a) * g[j - 1] + a * g[j]
            q0, q1 = g[j - 1, 1:5], g[j, 1:5]
            if np.dot(q0, q1) < 0:
                q1 = -q1
            q = (1 - a) * q0 + a * q1
            row[1:5] = q / np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])   # (sequential sum: the C++ harness does the same)
            row[0] = t
            out[i] = row
        return out


def write_mav0(mav_dir, frames, ts, imu, gt, K, dist=(0.0, 0.0, 0.0, 0.0)):
    """emit a stream in the EuRoC layout (the synthetic configs are 'emitted in EuRoC mav0 layout')."""
    from PIL import Image

    os.makedirs(os.path.join(mav_dir, "cam0", "data"), exist_ok=True)
    os.makedirs(os.path.join(mav_dir, "imu0"), exist_ok=True)
    os.makedirs(os.path.join(mav_dir, "state_groundtruth_estimate0"), exist_ok=True)
    h, w = frames[0].shape
    with open(os.path.join(mav_dir, "cam0", "sensor.yaml"), "w") as f:
        f.write(f"sensor_type: camera\nrate_hz: 20\nresolution: [{w}, {h}]\ncamera_model: pinhole\n"
                f"intrinsics: [{float(K[0, 0])!r}, {float(K[1, 1])!r}, {float(K[0, 2])!r}, {float(K[1, 2])!r}]\ndistortion_model: radial-tangential\n"
                f"distortion_coefficients: [{', '.join(repr(float(d)) for d in dist)}]\n")
    with open(os.path.join(mav_dir, "imu0", "sensor.yaml"), "w") as f:
        f.write("sensor_type: imu\nrate_hz: 200\n")
    ns = lambda t: int(round(t * 1e9))  # noqa: E731
    with open(os.path.join(mav_dir, "cam0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],filename\n")
        for t, img in zip(ts, frames):
            f.write(f"{ns(t)},{ns(t)}.png\n")
            Image.fromarray(img).save(os.path.join(mav_dir, "cam0", "data", f"{ns(t)}.png"))
    with open(os.path.join(mav_dir, "imu0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z\n")
        for r in imu:
            f.write(f"{ns(r[0])}," + ",".join(repr(float(v)) for v in r[1:7]) + "\n")
    with open(os.path.join(mav_dir, "state_groundtruth_estimate0", "data.csv"), "w") as f:
        f.write("#timestamp,p_x,p_y,p_z,q_w,q_x,q_y,q_z,v_x,v_y,v_z,bw_x,bw_y,bw_z,ba_x,ba_y,ba_z\n")
        for r in gt:  # gt rows: t, q(x,y,z,w), p, v, bg, ba
            vals = list(r[5:8]) + [r[4]] + list(r[1:4]) + list(r[8:17])
            f.write(f"{ns(r[0])}," + ",".join(repr(float(v)) for v in vals) + "\n")


# ------------------------------------------------------------------------------------------------ replay + evaluation
def replay(lib, handle, dataset, max_frames=None):
    """examples/test_euroc.cpp:46-95 over librdvio_pipeline.so.  Returns the trajectory rows (t, p(3), q(x,y,z,w)) of the
    newest tracked frame after every processed camera frame (Handler::get_latest_state) and the pipeline seconds."""
    import ctypes
    import time

    traj, spent, n_img = [], 0.0, 0
    pose = np.zeros(7)
    tt = ctypes.c_double(0)
    cnt = np.zeros(29, dtype=np.int64)
    seen = 0
    for clip in dataset.clips:
        if "gyro" in clip:
            acc, gyr = np.ascontiguousarray(clip["acc"]), np.ascontiguousarray(clip["gyro"])
            t0 = time.perf_counter()
            rc = lib.rdvio_pipeline_add_motion(handle, ctypes.c_double(clip["t"]), acc.ctypes.data_as(ctypes.c_void_p),
                                               gyr.ctypes.data_as(ctypes.c_void_p))
            spent += time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError(lib.rdvio_pipeline_last_error(handle).decode())
        if "image" in clip:
            if max_frames is not None and n_img >= max_frames:
                break
            img = np.ascontiguousarray(dataset.read_image(clip))
            t0 = time.perf_counter()
            rc = lib.rdvio_pipeline_add_frame(handle, ctypes.c_double(clip["t"]), img.ctypes.data_as(ctypes.c_void_p), img.shape[1],
                                              img.shape[0], img.shape[1], None)
            spent += time.perf_counter() - t0
            n_img += 1
            if rc != 0:
                raise RuntimeError(lib.rdvio_pipeline_last_error(handle).decode())
        lib.rdvio_pipeline_counters(handle, cnt.ctypes.data_as(ctypes.c_void_p))
        if cnt[0] != seen:
            seen = int(cnt[0])
            if lib.rdvio_pipeline_latest_state(handle, ctypes.byref(tt), pose.ctypes.data_as(ctypes.c_void_p)):
                traj.append([tt.value, pose[4], pose[5], pose[6], pose[0], pose[1], pose[2], pose[3]])
    return np.array(traj), spent


def write_tum(path, traj):
    with open(path, "w") as f:
        for r in traj:
            f.write(" ".join(f"{v:.9f}" for v in r) + "\n")


def ate_rmse(p_est, p_ref):
    """position RMSE after the best rigid (Umeyama without scale) alignment of p_est onto p_ref."""
    a, b = np.asarray(p_est), np.asarray(p_ref)
    ma, mb = a.mean(0), b.mean(0)
    H = (a - ma).T @ (b - mb)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return float(np.sqrt(np.mean(np.sum(((a - ma) @ R.T + mb - b) ** 2, axis=1))))
