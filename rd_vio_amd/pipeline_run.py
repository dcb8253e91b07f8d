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
        ("threading", ctypes.c_int32),
    ]


def load_pipeline_lib():
    from rd_vio_amd import build as rbuild
    rbuild.build()
    lib = ctypes.CDLL(rbuild.PIPE_LIB)
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
