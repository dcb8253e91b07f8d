"""Python driver of librdvio_pipeline.so over the HIP backend: feeds a stream the way the reference's test_euroc loop
does (examples/test_euroc.cpp:46-95) and returns the trajectory.  Product-side helper (bench.py, examples); the
oracle-backed twin used for the CPU-path comparison lives in tests/pipeline_util.py."""
import ctypes
import time

import numpy as np


class PipelineConfig(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("K", ctypes.c_double * 9),
        ("q_bc", ctypes.c_double * 4), ("p_bc", ctypes.c_double * 3),
        ("q_bi", ctypes.c_double * 4), ("p_bi", ctypes.c_double * 3),
        ("q_bo", ctypes.c_double * 4), ("p_bo", ctypes.c_double * 3),
        ("keypoint_noise_cov", ctypes.c_double * 4),
        ("gyroscope_noise_cov", ctypes.c_double * 9), ("accelerometer_noise_cov", ctypes.c_double * 9),
        ("gyroscope_bias_noise_cov", ctypes.c_double * 9), ("accelerometer_bias_noise_cov", ctypes.c_double * 9),
        ("sliding_window_size", ctypes.c_int32), ("sliding_window_subframe_size", ctypes.c_int32),
        ("sliding_window_force_keyframe_landmarks", ctypes.c_int32), ("sliding_window_tracker_frequent", ctypes.c_int32),
        ("feature_tracker_min_keypoint_distance", ctypes.c_double),
        ("feature_tracker_max_keypoint_detection", ctypes.c_int32), ("feature_tracker_max_init_frames", ctypes.c_int32),
        ("feature_tracker_max_frames", ctypes.c_int32),
        ("feature_tracker_clahe_clip_limit", ctypes.c_double),
        ("feature_tracker_clahe_width", ctypes.c_int32), ("feature_tracker_clahe_height", ctypes.c_int32),
        ("feature_tracker_predict_keypoints", ctypes.c_int32),
        ("initializer_keyframe_num", ctypes.c_int32), ("initializer_keyframe_gap", ctypes.c_int32),
        ("initializer_min_matches", ctypes.c_int32), ("initializer_min_triangulation", ctypes.c_int32),
        ("initializer_min_landmarks", ctypes.c_int32), ("initializer_min_parallax", ctypes.c_double),
        ("solver_iteration_limit", ctypes.c_int32),
        ("rotation_misalignment_threshold", ctypes.c_double), ("rotation_ransac_threshold", ctypes.c_double),
        ("random", ctypes.c_int32), ("parsac_flag", ctypes.c_int32), ("parsac_keyframe_check_size", ctypes.c_int32),
        ("threading", ctypes.c_int32), ("initializer_refine_imu", ctypes.c_int32), ("tracker_gates_on_backend", ctypes.c_int32),
    ]


def load_pipeline_lib(path=None):
    from rd_vio_amd import build as rbuild
    rbuild.build()
    lib = ctypes.CDLL(path or rbuild.PIPE_LIB)
    lib.rdvio_pipeline_last_error.restype = ctypes.c_char_p
    return lib


def default_config(lib, K, w, h, extr, noise, **over):
    cfg = PipelineConfig()
    lib.rdvio_pipeline_config_default(ctypes.byref(cfg))
    cfg.width, cfg.height = w, h
    cfg.K[:] = list(np.asarray(K, dtype=np.float64).ravel())
    cfg.q_bc[:] = list(extr[0:4])
    cfg.p_bc[:] = list(extr[4:7])
    cfg.q_bi[:] = list(extr[7:11])
    cfg.p_bi[:] = list(extr[11:14])
    cfg.keypoint_noise_cov[:] = [0.5, 0.0, 0.0, 0.5]      # configs/euroc_sensor.yaml keypoint noise
    cfg.gyroscope_noise_cov[:] = list(noise[0:9])
    cfg.accelerometer_noise_cov[:] = list(noise[9:18])
    cfg.gyroscope_bias_noise_cov[:] = list(noise[18:27])
    cfg.accelerometer_bias_noise_cov[:] = list(noise[27:36])
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def create_hip_pipeline(lib, ctx, cfg):
    """rdvio_pipeline_create_hip over an rd_vio_amd.Context; returns the pipeline handle."""
    h = ctypes.c_void_p()
    rc = lib.rdvio_pipeline_create_hip(ctypes.byref(h), ctypes.byref(cfg), ctx._h)
    if rc != 0:
        raise RuntimeError(f"rdvio_pipeline_create_hip failed ({rc})")
    return h


def feed_stream(lib, handle, frames, ts, imu, per_frame=None):
    """Push IMU samples and frames in timestamp order.  per_frame(k_processed) is called whenever the feature tracker has
    consumed another frame.  Returns wall-clock seconds spent inside the pipeline calls."""
    cnt = np.zeros(29, dtype=np.int64)
    seen = 0
    spent = 0.0
    ii = 0

    def push_imu(row):
        nonlocal spent, seen
        acc = np.ascontiguousarray(row[4:7])
        gyr = np.ascontiguousarray(row[1:4])
        t0 = time.perf_counter()
        rc = lib.rdvio_pipeline_add_motion(handle, ctypes.c_double(row[0]), acc.ctypes.data_as(ctypes.c_void_p),
                                           gyr.ctypes.data_as(ctypes.c_void_p))
        spent += time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(lib.rdvio_pipeline_last_error(handle).decode())
        if per_frame is not None:
            lib.rdvio_pipeline_counters(handle, cnt.ctypes.data_as(ctypes.c_void_p))
            if cnt[0] != seen:
                seen = int(cnt[0])
                per_frame(seen)

    for k, t in enumerate(ts):
        while ii < len(imu) and imu[ii, 0] <= t:
            push_imu(imu[ii])
            ii += 1
        img = np.ascontiguousarray(frames[k])
        t0 = time.perf_counter()
        rc = lib.rdvio_pipeline_add_frame(handle, ctypes.c_double(t), img.ctypes.data_as(ctypes.c_void_p), img.shape[1], img.shape[0],
                                          img.shape[1], None)
        spent += time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(lib.rdvio_pipeline_last_error(handle).decode())
    while ii < len(imu):
        push_imu(imu[ii])
        ii += 1
    return spent


class Replay(ctypes.Structure):
    """rdvio_replay (include/rdvio_pipeline.h)"""
    _fields_ = [("n_frames", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("stride", ctypes.c_int32),
                ("frames", ctypes.POINTER(ctypes.c_void_p)), ("frame_t", ctypes.c_void_p), ("n_imu", ctypes.c_int32), ("imu", ctypes.c_void_p),
                ("kp_capacity", ctypes.c_int32), ("kp_ids", ctypes.c_void_p), ("kp_xy", ctypes.c_void_p), ("kp_n", ctypes.c_void_p),
                ("latest", ctypes.c_void_p), ("window", ctypes.c_void_p), ("sys_state", ctypes.c_void_p), ("done_s", ctypes.c_void_p),
                ("frames_processed", ctypes.c_int32), ("elapsed_s", ctypes.c_double), ("flush", ctypes.c_int32), ("imu_consumed", ctypes.c_int32)]


def replay_stream(lib, handle, frames, ts, imu, kp_capacity=0, flush=2):
    """rdvio_pipeline_replay: a stream (or a segment of one: flush = 1, continue with imu[imu_consumed:]) in one native call (the
    test_euroc loop inside the library).  Returns a dict: window (rows x 17), latest (rows x 8), sys_state, done_s (seconds since the
    start of the call at which each frame was done), keypoints [(ids, xy)] when kp_capacity > 0, frames_processed, elapsed_s,
    imu_consumed.  Rows: one per frame the feature tracker consumed during the call."""
    n = len(ts)
    imgs = [np.ascontiguousarray(f) for f in frames]
    h, w = imgs[0].shape
    ptrs = (ctypes.c_void_p * n)(*[im.ctypes.data for im in imgs])
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    imu = np.ascontiguousarray(imu, dtype=np.float64)
    window, latest = np.zeros((n, 17)), np.zeros((n, 8))
    sys_state, done = np.zeros(n, dtype=np.int32), np.zeros(n)
    rp = Replay(n_frames=n, width=w, height=h, stride=w, frames=ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_void_p)), frame_t=ts.ctypes.data,
                n_imu=len(imu), imu=imu.ctypes.data, kp_capacity=kp_capacity, window=window.ctypes.data, latest=latest.ctypes.data,
                sys_state=sys_state.ctypes.data, done_s=done.ctypes.data, flush=flush)
    if kp_capacity > 0:
        ids = np.zeros((n, kp_capacity), dtype=np.int64)
        xy = np.zeros((n, kp_capacity, 2))
        kn = np.zeros(n, dtype=np.int32)
        rp.kp_ids, rp.kp_xy, rp.kp_n = ids.ctypes.data, xy.ctypes.data, kn.ctypes.data
    rc = lib.rdvio_pipeline_replay(handle, ctypes.byref(rp))
    if rc != 0:
        raise RuntimeError(lib.rdvio_pipeline_last_error(handle).decode())
    m = rp.frames_processed
    out = dict(window=window[:m], latest=latest[:m], sys_state=sys_state[:m], done_s=done[:m], frames_processed=m, elapsed_s=rp.elapsed_s,
               imu_consumed=int(rp.imu_consumed))
    if kp_capacity > 0:
        out["keypoints"] = [(ids[k, :min(kn[k], kp_capacity)].copy(), xy[k, :min(kn[k], kp_capacity)].copy()) for k in range(m)]
    return out



def config_dir():
    import os
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


def baseline_config(lib, window, features, sensor_yaml=None, setting_yaml=None, width=None, height=None, K=None, **extra):
    """The BASELINE configuration: configs/baseline_setting.yaml (the reference's shipped settings, restated) + the sensor
    calibration, with ONLY sliding_window.size and feature_tracker.max_keypoint_detection overridden;
    `extra` is for what is not a setting of the reference (threading).  A camera other than the EuRoC one (the 1280 x 720
    synthetic stream) replaces width / height / K.  Returns (PipelineConfig, dict of the values that were applied)."""
    import os

    from rd_vio_amd import euroc

    sensor_yaml = sensor_yaml or os.path.join(config_dir(), "synthetic_euroc_sensor.yaml")
    setting_yaml = setting_yaml or os.path.join(config_dir(), "baseline_setting.yaml")
    Kc, w, h, extr, noise, over = euroc.config_overrides(sensor_yaml, setting_yaml)
    if K is not None:
        Kc, w, h = np.asarray(K, dtype=np.float64), int(width), int(height)
    over = dict(over, sliding_window_size=int(window), feature_tracker_max_keypoint_detection=int(features), **extra)
    cfg = PipelineConfig()
    lib.rdvio_pipeline_config_default(ctypes.byref(cfg))
    cfg.width, cfg.height = w, h
    cfg.K[:] = list(np.asarray(Kc, dtype=np.float64).ravel())
    cfg.q_bc[:] = list(extr[0:4])
    cfg.p_bc[:] = list(extr[4:7])
    cfg.q_bi[:] = list(extr[7:11])
    cfg.p_bi[:] = list(extr[11:14])
    cfg.gyroscope_noise_cov[:] = list(noise[0:9])
    cfg.accelerometer_noise_cov[:] = list(noise[9:18])
    cfg.gyroscope_bias_noise_cov[:] = list(noise[18:27])
    cfg.accelerometer_bias_noise_cov[:] = list(noise[27:36])
    euroc.apply_overrides(cfg, over)
    return cfg, over


COUNTER_NAMES = ("preprocess", "detect", "track", "preintegrate", "ba_solve", "marginalize", "image_create")


def counters_report(cnt):
    """rdvio_pipeline_counters as a dict"""
    n = max(int(cnt[0]), 1)
    return {"frames": int(cnt[0]), "window_solves": int(cnt[1]), "marginalizations": int(cnt[3]), "localizations": int(cnt[4]),
            "subwindow_solves": int(cnt[5]), "largest_solve": {"frames": int(cnt[8]), "factors": int(cnt[9])},
            "solver_iterations": int(cnt[10]), "no_translation_frames": int(cnt[25]),
            "rd_path": {"imu_parsac_judgements": int(cnt[27]), "tracks_marked_dynamic": int(cnt[28])},
            "backend_ms_per_frame": {name: round(float(cnt[11 + 2 * k]) / 1e3 / n, 4) for k, name in enumerate(COUNTER_NAMES)},
            "backend_calls": {name: int(cnt[12 + 2 * k]) for k, name in enumerate(COUNTER_NAMES)}}


def run_pipeline(lib, make_pipeline, frames, ts, imu, init_states=None, kp_capacity=0):
    """Create a pipeline with make_pipeline(byref(handle)) -> rc, replay the stream natively, destroy it.  init_states: the
    optional bootstrap rows (rdvio_pipeline_set_init_states); None / empty = the full initializer.  Returns the replay dict
    plus `counters`."""
    h = ctypes.c_void_p()
    rc = make_pipeline(ctypes.byref(h))
    if rc != 0:
        raise RuntimeError(f"pipeline creation failed ({rc})")
    try:
        if init_states is not None and len(init_states):
            g = np.ascontiguousarray(init_states, dtype=np.float64)
            assert lib.rdvio_pipeline_set_init_states(h, len(g), g.ctypes.data_as(ctypes.c_void_p)) == 0
        out = replay_stream(lib, h, frames, ts, imu, kp_capacity=kp_capacity)
        cnt = np.zeros(29, dtype=np.int64)
        lib.rdvio_pipeline_counters(h, cnt.ctypes.data_as(ctypes.c_void_p))
        out["counters"] = cnt
        return out
    finally:
        lib.rdvio_pipeline_destroy(h)
