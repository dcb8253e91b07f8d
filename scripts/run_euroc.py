#!/usr/bin/env python3
"""Headless EuRoC replay on the HIP path (the reference's examples/test_euroc.cpp without its viewer / SlimeVR output):
  python scripts/run_euroc.py MAV0_DIR --sensor configs/euroc_sensor.yaml --setting configs/setting.yaml --out traj.txt
Writes a TUM-format trajectory and, when mav0/state_groundtruth_estimate0 exists, prints the ATE (after rigid alignment).
The window is initialised by the pipeline's own SfM + IMU-alignment initializer; --bootstrap-from-groundtruth replaces
those stages with the ground-truth states of the first keyframes."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mav_dir")
    ap.add_argument("--sensor", required=True, help="euroc_sensor.yaml (camera / IMU calibration)")
    ap.add_argument("--setting", default=None, help="setting.yaml (tracker / window / solver settings)")
    ap.add_argument("--out", default="trajectory_tum.txt")
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--bootstrap-from-groundtruth", action="store_true")
    args = ap.parse_args()

    import rd_vio_amd
    from rd_vio_amd import euroc
    from rd_vio_amd import pipeline_run as pr

    K, w, h, extr, noise, over = euroc.config_overrides(args.sensor, args.setting)
    ds = euroc.EurocDataset(args.mav_dir)
    if args.bootstrap_from_groundtruth and ds.groundtruth is None:
        raise SystemExit("--bootstrap-from-groundtruth needs mav0/state_groundtruth_estimate0/data.csv")
    lib = pr.load_pipeline_lib()
    cfg = euroc.apply_overrides(pr.default_config(lib, K, w, h, extr, noise), over)
    ctx = rd_vio_amd.Context(max_width=w, max_height=h, max_features=4096, max_window=max(16, cfg.sliding_window_size),
                             max_factors=40000)
    handle = pr.create_hip_pipeline(lib, ctx, cfg)
    frame_times = [c["t"] for c in ds.clips if "image" in c]
    if args.bootstrap_from_groundtruth:
        import ctypes

        init = np.ascontiguousarray(ds.init_states_at(frame_times[:cfg.initializer_keyframe_num * cfg.initializer_keyframe_gap * 4]))
        lib.rdvio_pipeline_set_init_states(handle, len(init), init.ctypes.data_as(ctypes.c_void_p))
    traj, spent = euroc.replay(lib, handle, ds, args.max_frames)
    lib.rdvio_pipeline_destroy(handle)
    ctx.close()
    euroc.write_tum(args.out, traj)
    report = {"frames": len(frame_times) if args.max_frames is None else min(args.max_frames, len(frame_times)),
              "poses": int(len(traj)), "pipeline_seconds": round(spent, 3), "trajectory": args.out}
    if len(traj) >= 3 and ds.groundtruth is not None:
        gt = ds.init_states_at(traj[:, 0])
        report["ate_rmse_m"] = round(euroc.ate_rmse(traj[:, 1:4], gt[:, 5:8]), 5)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
