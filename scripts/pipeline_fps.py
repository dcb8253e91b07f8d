#!/usr/bin/env python3
"""Frames/s of the product pipeline (librdvio_pipeline.so over the HIP backend) on a synthetic EuRoC-shaped stream with the
BASELINE configuration, per threading mode -- the development twin of bench.py's headline leg.
  python scripts/pipeline_fps.py [--frames 400] [--window 8] [--features 150] [--modes 0,2] [--bootstrap init|gt] [--cpu]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--window", type=int, default=8)
    ap.add_argument("--features", type=int, default=150)
    ap.add_argument("--modes", default="0,2")
    ap.add_argument("--bootstrap", default="init", choices=("init", "gt"))
    ap.add_argument("--mover", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="also run the CPU path (oracle backend, threading 1) and compare")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--hd", action="store_true", help="the 1280 x 720 camera of BASELINE config 5 (focal 900) instead of the EuRoC one")
    ap.add_argument("--gates", type=int, default=0, choices=(0, 1), help="1: the tracker's two-view gates + thinning behind the backend hooks")
    args = ap.parse_args()

    import rd_vio_amd
    from rd_vio_amd import pipeline_run as pr
    from rd_vio_amd import synth

    W, H, K = 752, 480, synth.EUROC_K
    kw = {}
    if args.hd:
        W, H = 1280, 720
        K = np.array([[900.0, 0, 640.0], [0, 900.0, 360.0], [0, 0, 1.0]])
        kw = dict(width=W, height=H, K=K)
    t0 = time.time()
    frames, ts, imu, gt = synth.make_stream(args.frames, W, H, K, mover=args.mover)
    print(f"# rendered {args.frames} frames in {time.time() - t0:.1f} s", file=sys.stderr)
    lib = pr.load_pipeline_lib()
    init = gt if args.bootstrap == "gt" else None
    res = {}
    runs = []
    for mode in [int(m) for m in args.modes.split(",")]:
        for rep in range(args.repeat):
            cfg, over = pr.baseline_config(lib, args.window, args.features, threading=mode, tracker_gates_on_backend=args.gates, **kw)
            ctx = rd_vio_amd.Context(max_width=W, max_height=H, max_features=max(1024, 4 * args.features), max_window=args.window + 8, max_factors=40000)
            try:
                r = pr.run_pipeline(lib, lambda out: lib.rdvio_pipeline_create_hip(out, __import__("ctypes").byref(cfg), ctx._h), frames, ts, imu, init, kp_capacity=2048)
            finally:
                ctx.close()
            st = r["window"]
            ok = ~np.isnan(st[:, 0])
            i0 = int(np.argmax(ok))
            d = r["done_s"]
            rep_ = pr.counters_report(r["counters"])
            rep_.update(threading=mode, fps_all=round(len(d) / r["elapsed_s"], 1), first_tracking_frame=i0,
                        fps_tracking=round((len(d) - 1 - i0) / (d[-1] - d[i0]), 1), ms_per_frame_tracking=round(1e3 * (d[-1] - d[i0]) / (len(d) - 1 - i0), 4))
            rep_["fps_per_100_frames"] = [round(100.0 / (d[k + 100] - d[k]), 1) for k in range(i0, len(d) - 101, 100)]
            res[mode] = (r, rep_)
            runs.append((mode, r))
            print(json.dumps(rep_))
    if args.cpu:
        import oracle

        shim = oracle.build_backend()
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pipeline_util as pu

        cfg, over = pr.baseline_config(lib, args.window, args.features, threading=1, **kw)
        c = pr.run_pipeline(lib, pu.oracle_pipeline_factory(lib, shim, cfg), frames, ts, imu, init, kp_capacity=2048)
        for mode, r in runs:
            if mode == 0:
                continue
            same = len(r["keypoints"]) == len(c["keypoints"]) and all(np.array_equal(a[0], b[0]) for a, b in zip(r["keypoints"], c["keypoints"]))
            samexy = same and all(np.array_equal(a[1], b[1]) for a, b in zip(r["keypoints"], c["keypoints"]))
            sg, sc = r["window"], c["window"]
            both = ~np.isnan(sg[:, 0]) & ~np.isnan(sc[:, 0])
            print(json.dumps({"vs_cpu_path": mode, "indices_identical": bool(same), "pixels_identical": bool(samexy),
                              "max_pos_diff_mm": float(1e3 * np.abs(sg[both, 5:8] - sc[both, 5:8]).max()),
                              "cpu_fps": round(len(c["done_s"]) / c["elapsed_s"], 1), "counters_equal": bool((r["counters"][:11] == c["counters"][:11]).all()),
                              "first_frame_with_different_indices": next((k for k, (a, b) in enumerate(zip(r["keypoints"], c["keypoints"])) if not np.array_equal(a[0], b[0])), None),
                              "first_frame_with_different_state": next((k for k in range(min(len(sg), len(sc))) if both[k] and not np.array_equal(sg[k], sc[k])), None)}))


if __name__ == "__main__":
    main()
