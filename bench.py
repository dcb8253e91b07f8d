#!/usr/bin/env python3
"""bench.py -- frames/sec of rd_vio's per-frame hot path on MI355X (BASELINE.json metric).

`value`: camera frames per second of the PRODUCT PIPELINE (librdvio_pipeline.so over librdvio_hip.so: feature tracker on the
caller's thread, sliding-window estimator on the worker thread -- rdvio_pipeline_config::threading = 2) in its tracking phase,
fed like test_euroc feeds rdvio::Odometry (IMU samples and camera frames in timestamp order, rdvio_pipeline_replay) with a
synthetic EuRoC-shaped stream (there is no EuRoC data on the box) that is resident in host memory, on the BASELINE
configuration: configs/baseline_setting.yaml (the reference's shipped settings, restated) with only sliding_window.size and
feature_tracker.max_keypoint_detection overridden, bootstrapped by the full initializer.  A "step" is one camera frame: LK
tracker + two-view gates + detection (A1-A6), preintegration (A7), RD dynamic-outlier path (A19), localisation / window /
subwindow solves and marginalisation (A8-A16).  The W warm-up frames and the K timed frames are tracking-phase frames; the
bootstrap frames before them are not timed.  This rate is PCIe-inclusive (host images in, host-built BA graphs in, states out).

Beside it: `kernel_loop` (the resident per-frame kernel sequence of rounds 1-2, every input in HBM, with per-stage HIP-event
times and the image kernels' HBM rooflines), `roofline` (dominant kernel ba_solve_kernel: algorithmic flops per launch over its
HIP-event duration inside the timed region), `cpu_baseline` (the same orchestration over the CPU oracle backend on the same
stream, one pinned core, this box), `end_to_end` (feature-index / trajectory agreement of the two paths).
Multi-GPU = one independent replica (one stream) per GPU, no collective on the data path; torch.distributed is
used only for the barrier / max-over-ranks timing.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1]: EuRoC V1_01_easy shape, 150 features, window 8
    "euroc_v101": dict(width=752, height=480, features=150, window=8, landmarks=150, iters=30,
                       name="EuRoC V1_01_easy-shaped synthetic stream 752x480, 150 features, window 8, LK + BA on GPU"),
    # configs[2]: MH_03_medium shape, 300 features, window 10, RD dynamic-outlier path on (end-to-end leg: a mapped object
    # starts to move, parsac_flag = 1)
    "euroc_mh03_rd": dict(width=752, height=480, features=300, window=10, landmarks=300, iters=30, parsac_stream=True,
                          name="EuRoC MH_03_medium-shaped synthetic stream 752x480, 300 features, window 10, RD path"),
    # configs[4]: roofline run
    # (bootstrapped with supplied keyframe states: on this stream the initializer's 8-keyframe SfM + IMU alignment ends in a scale
    # the tracker does not recover from -- on the CPU path and the GPU path alike, 100 m off after 250 frames --, and a diverging
    # estimator says nothing about either path; the initializer-bootstrapped rate is reported under pipeline_variants)
    "synthetic_720p": dict(width=1280, height=720, features=1000, window=16, landmarks=1000, iters=30, bootstrap="groundtruth",
                           name="synthetic 1280x720 stream, 1000 features, window 16"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector/matrix peak


def lk_algorithmic_bytes(n):
    """per feature.level.direction 22x22 B template + 22x22x4 B gradients + 22x22 B search patch
    (LDS-staged, read once) = 2904 B; 4 levels x 2 directions; + 17 B of point I/O per feature."""
    return n * 4 * 2 * (22 * 22 * (1 + 4 + 1)) + n * 17


def preprocess_algorithmic_bytes(L):
    """P0 (1 read + 1 write CLAHE) + sum_l P_l (1 read + 4 write deriv) + sum_{l>=1} P_l write."""
    P = [L.w[i] * L.h[i] for i in range(L.levels)]
    return 2 * P[0] + 5 * sum(P) + sum(P[1:])


def ba_algorithmic_flops(pb, iterations, successful_steps):
    """FP64 flops of one ba_solve_kernel launch from SURVEY.md 8(d)'s per-unit figures: per linearisation
    0.5 kflop per reprojection factor (A8) + 15 kflop per preintegration factor (A11) + 2.D^2.16W for the prior's
    Jacobian product (A12) + 182 MAC per factor and 36.m^2 MAC per landmark with m observations (normal equations +
    landmark Schur) + n^3/3 for the reduced Cholesky (n = 15 x free frames); per trial-step cost evaluation the
    residual halves only (0.2 kflop per reprojection factor, 2 kflop per preintegration factor, 2.D^2 for the
    prior).  Linearisations = successful steps + 1, cost evaluations = iterations."""
    F = len(pb["tgt"])
    P = len(pb.get("pre_i", ()))
    Wp = len(pb["prior_frames"]) if "prior_frames" in pb else 0
    D = 15 * Wp
    nfree = int((np.asarray(pb["frame_fixed"]) == 0).sum())
    lm_free = np.asarray(pb["lm_fixed"]) == 0
    m = np.bincount(np.asarray(pb["lm"]), minlength=len(pb["inv_depth"]))[lm_free]
    n = 15 * nfree
    lin = 500.0 * F + 15000.0 * P + 2.0 * D * D * 16 * Wp + 2 * 182.0 * F + 2 * 36.0 * float((m * m).sum()) + n ** 3 / 3.0
    cost = 200.0 * F + 2000.0 * P + 2.0 * D * D
    return (successful_steps + 1) * lin + iterations * cost


PMC_TAG = "r03"


def kernel_source_hash():
    """sha256 over the HIP library's sources (rd_vio_amd/csrc/*, include/rdvio_hip.h): the PMC summary under profiles/ is only
    valid for the kernels it was taken from."""
    import hashlib

    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "rd_vio_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp", ".h")):
            hsh.update(f.encode())
            hsh.update(open(os.path.join(d, f), "rb").read())
    hsh.update(open(os.path.join(ROOT, "include", "rdvio_hip.h"), "rb").read())
    return hsh.hexdigest()


def pmc_traffic_bytes():
    """HBM-side bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate
    runs of this script, scripts/refresh_profiles.sh -> profiles/<tag>_pmc_fetch_write.csv + .meta.json).  Counters cannot be
    collected inside a timed run, so the figure is looked up -- but only while the kernel sources still hash to what the
    summary was taken from; otherwise traffic is reported as null with the reason.  FETCH_SIZE + WRITE_SIZE in KB, RAW: the
    gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md is calibrated for 16-B-per-lane streaming reads only; these
    kernels read bytes / shorts / scattered doubles (widths the guide calls uncalibrated), so the figure is a LOWER BOUND."""
    path = os.path.join(ROOT, "profiles", f"{PMC_TAG}_pmc_fetch_write.csv")
    meta = os.path.join(ROOT, "profiles", f"{PMC_TAG}_pmc_fetch_write.meta.json")
    if not (os.path.exists(path) and os.path.exists(meta)):
        return {}, f"no PMC summary profiles/{PMC_TAG}_pmc_fetch_write.csv for this round yet"
    if json.load(open(meta)).get("kernel_source_sha256") != kernel_source_hash():
        return {}, f"stale: profiles/{PMC_TAG}_pmc_fetch_write.csv was taken from other kernel sources (scripts/refresh_profiles.sh)"
    out = {}
    for line in open(path).read().strip().splitlines()[1:]:
        k, _n, f, w = line.split(",")
        out[k] = int((float(f) + float(w)) * 1024)
    return out, f"bytes per launch, FETCH_SIZE + WRITE_SIZE raw = lower bound (profiles/{PMC_TAG}_pmc_fetch_write.csv, sources verified by hash)"


def build_workload(cfg, ctx, torch, dev, seed=648, host=None):
    """Synthetic per-frame inputs, resident on the device.  `host`: the workload of another context of the same config --
    its host-side data (rendered frames, keypoints, IMU segments, BA problems) is reused and only the device side (frame
    slot 1 preprocessed, uploads into THIS context's arenas) is built: the sequences of the multi-sequence leg."""
    import rd_vio_amd
    from rd_vio_amd import synth

    w, h, nfeat, W = cfg["width"], cfg["height"], cfg["features"], cfg["window"]
    wl = {}
    # a short loop of frames: the same planar scene under a small per-frame motion
    offs = [(0.0, 0.0), (2.6, -1.4), (5.1, -2.9), (2.4, -1.2)]
    frames = host["frames_host"] if host else [synth.render_scene(w, h, seed=seed, offset=o, rot=0.002 * i) for i, o in enumerate(offs)]
    wl["frames_host"] = frames
    wl["frames"] = [torch.from_numpy(f).to(dev) for f in frames]
    # features: what detect_keypoints finds on frame 0 (min distance as in configs/setting.yaml:12)
    img0 = rd_vio_amd.HipImage(ctx, 1, frames[0])
    img0.preprocess()
    if host:
        kp = host["kp_host"]
    else:
        kp = img0.detect_keypoints(np.zeros((0, 2)), nfeat, 10.0)
        if len(kp) < nfeat:  # top up from a jittered grid so the feature count matches the config
            extra = synth.jittered_grid(w, h, 40, 25, seed=seed)[: nfeat - len(kp)]
            kp = np.concatenate([kp, extra])
        kp = kp[:nfeat]
    wl["kp_host"] = kp
    wl["curr"] = torch.from_numpy(kp.copy()).to(dev)
    wl["next"] = torch.zeros_like(wl["curr"])
    wl["status"] = torch.zeros(nfeat, dtype=torch.uint8, device=dev)
    # IMU: one frame segment (no covariance: feature_tracker.cpp:82-84) + W keyframe segments with covariance and
    # Jacobians (sliding_window_tracker.cpp:294)
    rng = np.random.default_rng(seed + 1)
    segs, par = [], []
    if host:
        segs, par = host["imu_segs"], [list(p) for p in host["imu_par"]]
    else:
        segs.append(synth.make_imu_segment(1.0, 1.05, rng=rng, bg=synth.TRUE_BG, ba=synth.TRUE_BA))
        par.append([1.05, *synth.TRUE_BG, *synth.TRUE_BA])
        for j in range(W):
            t0 = 1.0 + 0.25 * j
            segs.append(synth.make_imu_segment(t0, t0 + 0.25, rng=rng, bg=synth.TRUE_BG, ba=synth.TRUE_BA))
            par.append([t0 + 0.25, *synth.TRUE_BG, *synth.TRUE_BA])
    off = np.zeros(len(segs) + 1, dtype=np.int32)
    for i, s in enumerate(segs):
        off[i + 1] = off[i] + len(s)
    wl["imu_segs"], wl["imu_par"] = segs, np.array(par)
    wl["imu"] = torch.from_numpy(np.concatenate(segs)).to(dev)
    wl["imu_off"] = torch.from_numpy(off).to(dev)
    wl["imu_par_dev"] = torch.from_numpy(np.array(par)).to(dev)
    wl["noise"] = torch.from_numpy(synth.EUROC_NOISE.copy()).to(dev)
    wl["pre_out"] = torch.zeros((len(segs), rd_vio_amd.PREINT_SIZE), dtype=torch.float64, device=dev)
    # BA problems (preintegration through the HIP PreIntegrator, not the oracle)
    gpu_pre = lambda imu, t, bg, ba: ctx.preintegrate([imu], [t], [bg], [ba], synth.EUROC_NOISE)[0]  # noqa: E731
    K = synth.EUROC_K.copy()
    if (w, h) != (752, 480):
        K = np.array([[900.0, 0, w / 2.0], [0, 900.0, h / 2.0], [0, 0, 1.0]])
    if host:
        pb, loc = host["window_pb"], host["localize_pb"]
    else:
        pb = synth.make_window_problem(W + 1, cfg["landmarks"], seed, preintegrate=gpu_pre, K=K)
        loc = dict(pb)
        loc["frame_fixed"] = np.ones(W + 1, dtype=np.uint8)
        loc["frame_fixed"][W] = 0
        loc["lm_fixed"] = np.ones(len(pb["inv_depth"]), dtype=np.uint8)
        keep = pb["tgt"] == W
        for k in ("tgt", "ref", "lm", "tangent"):
            loc[k] = pb[k][keep]
        loc["pre_i"], loc["pre_j"], loc["preint"] = pb["pre_i"][-1:], pb["pre_j"][-1:], pb["preint"][-1:]
        for k in ("prior_frames", "lin", "S", "f"):
            loc.pop(k, None)
    wl["window_pb"] = pb
    wl["localize_pb"] = loc
    ctx.ba_upload(pb, slot=0)
    ctx.ba_upload(loc, slot=1)
    # marginalisation input: the steady-state prior (what the previous marginalisation leaves behind), as on every
    # marginalisation of a session but the first
    wl["marg_args"] = host["marg_args"] if host else synth.steady_state_marg_inputs(pb, ctx.marginalize)
    ctx.marginalize_upload(*wl["marg_args"])
    wl["L"] = img0.L
    return wl


class Sequence:
    """One VIO sequence on the device: its own rdvio_hip context (arenas, staging, three lanes = three HIP streams), the
    synthetic per-frame inputs resident in HBM, and step(k) = one camera frame of the hot path."""

    STAGES = ["preprocess", "lk_track", "detect", "preintegrate", "ba_localize", "ba_window", "marginalize"]
    # event pairs per stage: (start, end) indices into the per-step event list
    EV_PAIRS = [(0, 1), (1, 2), (2, 3), (3, 4), (5, 6), (7, 8), (9, 10)]
    N_EV = 11

    def __init__(self, cfg, torch, dev, device_index, overlap=True, first_stream=None, host=None, max_factors=20000):
        import ctypes

        import rd_vio_amd

        self.cfg, self.overlap, self.ctypes, self.rd = cfg, overlap, ctypes, rd_vio_amd
        # explicit (non-default) streams: the context enqueues on them and torch.cuda.Event records on them, so the HIP events
        # bracket exactly the kernels of each stage.  Lanes (include/rdvio_hip.h): the frontend (image side +
        # preintegration), the solver and the marginalisation each get their own stream unless serial.
        stream = first_stream if first_stream is not None else torch.cuda.Stream(device=dev)
        self.ctx = ctx = rd_vio_amd.Context(max_width=cfg["width"], max_height=cfg["height"], max_features=max(1024, 4 * cfg["features"]),
                                            max_window=cfg["window"], max_factors=max_factors, device=device_index, stream=stream.cuda_stream)
        self.s_front = self.s_solve = self.s_marg = stream
        if overlap:
            self.s_solve, self.s_marg = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            ctx.set_lane_stream(rd_vio_amd.binding.LANE_SOLVER, self.s_solve.cuda_stream)
            ctx.set_lane_stream(rd_vio_amd.binding.LANE_MARG, self.s_marg.cuda_stream)
        with torch.cuda.stream(stream):
            self.wl = build_workload(cfg, ctx, torch, dev, host=host)
        self.kp_buf = np.zeros((2 * cfg["features"], 2))
        self.n_out = ctypes.c_int(0)
        self.nseg = len(self.wl["imu_segs"])

    def frame_step_desc(self):
        """this sequence as a rdvio_frame_step (include/rdvio_hip.h): the same per-frame step as step(), for the native driver"""
        from rd_vio_amd.binding import FrameStep

        wl, cfg, ct = self.wl, self.cfg, self.ctypes
        self._images = (ct.c_void_p * len(wl["frames"]))(*[f.data_ptr() for f in wl["frames"]])
        return FrameStep(ctx=self.ctx._h, width=cfg["width"], height=cfg["height"], stride=cfg["width"], n_images=len(wl["frames"]),
                         images_dev=ct.cast(self._images, ct.POINTER(ct.c_void_p)), n_features=cfg["features"], keypoints_capacity=len(self.kp_buf),
                         curr_xy_dev=wl["curr"].data_ptr(), next_xy_dev=wl["next"].data_ptr(), status_dev=wl["status"].data_ptr(),
                         keypoints_host=self.kp_buf.ctypes.data, min_distance=10.0, nseg=self.nseg, ba_iterations=cfg["iters"],
                         seg_off_dev=wl["imu_off"].data_ptr(), imu_dev=wl["imu"].data_ptr(), par_dev=wl["imu_par_dev"].data_ptr(),
                         noise_dev=wl["noise"].data_ptr(), preint_out_dev=wl["pre_out"].data_ptr(), overlap=1 if self.overlap else 0, reserved=0)

    def estimator(self, ev):
        """frame k's estimation on the solver / marginalisation lanes: localize_newframe, refine_window (reads the prior the
        previous marginalisation wrote: device-side wait on the marginalisation lane), then slide_window ->
        Map::marginalize_frame(0) (reads the window solve's states: device-side wait on the solver lane)."""
        ctx, lib, h, iters = self.ctx, self.ctx._lib, self.ctx._h, self.cfg["iters"]
        LS, LM = self.rd.binding.LANE_SOLVER, self.rd.binding.LANE_MARG
        if ev: ev[5].record(self.s_solve)
        ctx._check(lib.rdvio_hip_ba_solve_resident(h, 1, iters))
        if ev: ev[6].record(self.s_solve)
        ctx.lane_wait(LS, LM)
        if ev: ev[7].record(self.s_solve)
        ctx._check(lib.rdvio_hip_ba_solve_resident(h, 0, iters))
        if ev: ev[8].record(self.s_solve)
        ctx.lane_wait(LM, LS)
        if ev: ev[9].record(self.s_marg)
        ctx._check(lib.rdvio_hip_marginalize_resident(h, 0))
        if ev: ev[10].record(self.s_marg)

    def frontend(self, k, ev):
        ctx, lib, h, wl, cfg = self.ctx, self.ctx._lib, self.ctx._h, self.wl, self.cfg
        w, hh, nfeat = cfg["width"], cfg["height"], cfg["features"]
        cur, prv = k % 2, (k + 1) % 2
        img = wl["frames"][k % len(wl["frames"])]
        s_front = self.s_front
        if ev: ev[0].record(s_front)
        ctx._check(lib.rdvio_hip_image_preprocess_dev(h, cur, img.data_ptr(), w, hh, w, 6.0, 8, 8))
        if ev: ev[1].record(s_front)
        ctx._check(lib.rdvio_hip_track_keypoints_dev(h, prv, cur, nfeat, wl["curr"].data_ptr(), wl["next"].data_ptr(),
                                                     0, wl["status"].data_ptr()))
        if ev: ev[2].record(s_front)
        ctx._check(lib.rdvio_hip_detect_keypoints(h, cur, self.kp_buf.ctypes.data, 0, len(self.kp_buf), nfeat, 10.0,
                                                  self.ctypes.byref(self.n_out)))
        if ev: ev[3].record(s_front)
        # frame segment without covariance, keyframe segments with (two launches, as the reference's two call sites)
        ctx._check(lib.rdvio_hip_preintegrate_dev(h, 1, wl["imu_off"].data_ptr(), wl["imu"].data_ptr(),
                                                  wl["imu_par_dev"].data_ptr(), wl["noise"].data_ptr(), 0, 0,
                                                  wl["pre_out"].data_ptr()))
        ctx._check(lib.rdvio_hip_preintegrate_dev(h, self.nseg - 1, wl["imu_off"].data_ptr() + 4, wl["imu"].data_ptr(),
                                                  wl["imu_par_dev"].data_ptr() + 56, wl["noise"].data_ptr(), 1, 1,
                                                  wl["pre_out"].data_ptr() + 8 * self.rd.PREINT_SIZE))
        if ev: ev[4].record(s_front)

    def step(self, k, ev=None):
        """One camera frame.  Overlapped (default): while the solver lane works on frame k's localisation and window solve
        and the marginalisation lane on its slide_window, the frontend lane runs frame k+1's image side and preintegration
        -- the reference's tracker-thread / frontend-thread split (handler.cpp:35-50).  The host waits once per frame for what
        host logic consumes before the next frame: the frontend's keypoints and the solver's states; the new prior stays on
        the device for the next window solve.  Serial: every stage back to back on one stream (round 1's step)."""
        if self.overlap:
            self.estimator(ev)
            self.frontend(k, ev)
            self.ctx.lane_sync(self.rd.binding.LANE_FRONTEND)
            self.ctx.lane_sync(self.rd.binding.LANE_SOLVER)
        else:
            self.frontend(k, ev)
            self.estimator(ev)
            self.ctx.sync()


def multi_sequence(cfg, torch, dev, device_index, host, n_seq, steps, warmup, overlap=False, wait_mode=-1):
    """n_seq independent sequences on ONE GPU, each with its own context and stream(s), each driven by its own host thread
    through the native driver (rdvio_hip_run_sequences: the same per-frame step as `value`, as one C call per frame).  One
    sequence occupies one compute unit for most of a frame (the persistent single-workgroup solver), so a GPU has room for
    many: this is what the device sustains when it is kept busy.  NOT `value` (BASELINE's configs are one stream per GPU):
    reported beside it."""
    import ctypes

    from rd_vio_amd.binding import FrameStep

    mf = max(2048, 2 * len(host["window_pb"]["tgt"]))
    seqs = [Sequence(cfg, torch, dev, device_index, overlap=overlap, host=host, max_factors=mf) for _ in range(n_seq)]
    torch.cuda.synchronize()
    descs = (FrameStep * n_seq)(*[sq.frame_step_desc() for sq in seqs])
    elapsed = ctypes.c_double(0.0)
    per = np.zeros(n_seq)
    rc = seqs[0].ctx._lib.rdvio_hip_run_sequences(descs, n_seq, warmup, steps, wait_mode, ctypes.byref(elapsed), per.ctypes.data)
    sm = [sq.ctx.ba_fetch(0)[2] for sq in seqs] if rc == 0 else []
    for sq in seqs:
        sq.ctx.close()
    if rc != 0:
        return {"sequences": n_seq, "error": f"rdvio_hip_run_sequences returned {rc}"}
    return {"sequences": n_seq, "streams_per_sequence": 3 if overlap else 1, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "host_wait": {-1: "auto", 0: "spin", 1: "block"}[wait_mode], "host_cores": os.cpu_count(),
            "frames_per_sequence": steps, "aggregate_fps": round(n_seq * steps / elapsed.value, 1),
            "per_sequence_fps": round(steps / elapsed.value, 1), "ms_per_frame_per_sequence": round(1e3 * elapsed.value / steps, 4),
            "slowest_over_fastest_sequence": round(float(per.max() / per.min()), 3),
            "window_solve_iterations": sorted({int(m.iterations) for m in sm}),
            "note": "independent sequences sharing one GPU (one context and one host thread each, rdvio_hip_run_sequences); the same "
                    "per-frame step as `value`; wall time from the common start to the last sequence's last frame"}


BOOT_FRAMES = 44   # the initializer (8 keyframes, gap 5) hands over at frame 36 of the synthetic stream; margin for other cameras


def make_stream(cfg, n_frames):
    from rd_vio_amd import synth

    w, h = cfg["width"], cfg["height"]
    K = synth.EUROC_K.copy()
    if (w, h) != (752, 480):
        K = np.array([[900.0, 0, w / 2.0], [0, 900.0, h / 2.0], [0, 0, 1.0]])
    frames, ts, imu, gt = synth.make_stream(n_frames, w, h, K, mover=bool(cfg.get("parsac_stream")))
    return dict(frames=frames, ts=np.ascontiguousarray(ts), imu=np.ascontiguousarray(imu), gt=np.ascontiguousarray(gt), K=K, w=w, h=h)


def pipeline_config(lib, cfg, threading, **extra):
    from rd_vio_amd import pipeline_run as pr

    kw = dict(extra)
    if (cfg["width"], cfg["height"]) != (752, 480):
        kw.update(width=cfg["width"], height=cfg["height"], K=np.array([[900.0, 0, cfg["width"] / 2.0], [0, 900.0, cfg["height"] / 2.0], [0, 0, 1.0]]))
    return pr.baseline_config(lib, cfg["window"], cfg["features"], threading=threading, **kw)


class PipelineRun:
    """One pipeline over one stream, replayed in segments: bootstrap + warm-up (untimed), then the timed frames."""

    def __init__(self, lib, make_pipeline, stream, init_states=None, kp_capacity=0):
        import ctypes

        from rd_vio_amd import pipeline_run as pr

        self.lib, self.pr, self.stream, self.kp_capacity, self.ct = lib, pr, stream, kp_capacity, ctypes
        self.h = ctypes.c_void_p()
        rc = make_pipeline(ctypes.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"pipeline creation failed ({rc})")
        if init_states is not None and len(init_states):
            g = np.ascontiguousarray(init_states, dtype=np.float64)
            assert lib.rdvio_pipeline_set_init_states(self.h, len(g), g.ctypes.data_as(ctypes.c_void_p)) == 0
        self.frame0 = 0
        self.imu0 = 0
        self.rows = []

    def segment(self, n, last=False):
        s = self.stream
        a, b = self.frame0, self.frame0 + n
        out = self.pr.replay_stream(self.lib, self.h, s["frames"][a:b], s["ts"][a:b], s["imu"][self.imu0:], kp_capacity=self.kp_capacity,
                                    flush=2 if last else 1)
        if self.lib.rdvio_pipeline_drain(self.h) != 0:
            raise RuntimeError(self.lib.rdvio_pipeline_last_error(self.h).decode())
        self.frame0, self.imu0 = b, self.imu0 + out["imu_consumed"]
        self.rows.append(out)
        return out

    def counters(self):
        cnt = np.zeros(29, dtype=np.int64)
        self.lib.rdvio_pipeline_counters(self.h, cnt.ctypes.data_as(self.ct.c_void_p))
        return cnt

    def close(self):
        if self.h:
            self.lib.rdvio_pipeline_destroy(self.h)
            self.h = None

    def table(self, key):
        return np.concatenate([r[key] for r in self.rows]) if self.rows else np.zeros((0,))

    def keypoints(self):
        return [kp for r in self.rows for kp in r.get("keypoints", [])]


def bootstrap_and_warm_up(run, warmup, n_total):
    """replays until the pipeline has been tracking for `warmup` frames; returns the number of frames consumed"""
    run.segment(BOOT_FRAMES)
    while True:
        st = run.table("sys_state")
        tracking = int((st == 1).sum())
        if tracking >= warmup:
            return run.frame0
        more = max(warmup - tracking, 1) if tracking > 0 else 8
        if run.frame0 + more > n_total:
            raise RuntimeError(f"the pipeline did not reach its tracking phase within {run.frame0} frames")
        run.segment(more)


def cpu_path_worker(args):
    """child process of cpu_baseline(): the orchestration (the pipeline library build named on the command line) over the CPU
    oracle backend (the oracle build named by RDVIO_ORACLE_VARIANT), threading = 1 (the pipelined schedule on ONE thread), on
    the pickled stream, segmented exactly like the GPU run.  Prints one JSON line, saves the per-frame tables.  Never touches
    the GPU."""
    import pickle

    import oracle
    from rd_vio_amd import pipeline_run as pr

    with open(args.cpu_path_worker, "rb") as fh:
        cfg, stream, n_pre, steps, variant, lib_path = pickle.load(fh)
    lib = pr.load_pipeline_lib(lib_path)
    shim = oracle.build_backend(variant)
    pcfg, _ = pipeline_config(lib, cfg, threading=1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes

    class Backend(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in ("user", "image_create", "image_preprocess", "image_detect", "image_track", "image_release",
                                                   "image_destroy", "preintegrate", "ba_solve", "marginalize", "last_error", "destroy", "parsac_score",
                                                   "parsac_fetch", "preintegrate_estimator", "thread_attach", "marginalize_begin", "marginalize_end", "ransac_generate_score", "ransac_fetch", "thin_tracks", "parsac_generate_score", "preintegrate_estimator_begin", "preintegrate_estimator_end", "ba_solve_begin", "ba_solve_end")]

    be = Backend()
    shim.rdvio_oracle_backend_fill(ctypes.byref(be))
    run = PipelineRun(lib, lambda out: lib.rdvio_pipeline_create(out, ctypes.byref(pcfg), ctypes.byref(be)), stream,
                      stream["gt"] if cfg.get("bootstrap") == "groundtruth" else None, kp_capacity=2048)
    run.segment(n_pre)
    c0 = run.counters()
    t = run.segment(steps)
    c1 = run.counters()
    names = ("preprocess", "detect", "track", "preintegrate", "ba_solve", "marginalize", "image_create")
    stage = {n: round(float(c1[11 + 2 * k] - c0[11 + 2 * k]) / 1e3 / steps, 4) for k, n in enumerate(names)}
    np.savez(args.cpu_path_out, window=run.table("window"), kp_n=np.array([len(k[0]) for k in run.keypoints()]),
             kp_ids=np.concatenate([k[0] for k in run.keypoints()] + [np.zeros(0, dtype=np.int64)]),
             kp_xy=np.concatenate([k[1].reshape(-1, 2) for k in run.keypoints()] + [np.zeros((0, 2))]), counters=c1)
    run.close()
    print(json.dumps({"frames": steps, "seconds": t["elapsed_s"], "backend_ms_per_frame": stage}))


CPU_BUILDS = {
    # strong baseline: what the >= 200x target is quoted against
    "O3_native": (["-O3", "-march=native"], "O3_native"),
    # the reference's own flags, /root/reference/CMakeLists.txt:10,17-18
    "Og_fastmath": (["-Og", "-msse", "-msse2", "-msse3", "-mtune=native"], "Og_fastmath"),
    # the parity checker's build (what the trajectory / feature-index comparison runs on)
    "O2_parity": (None, None),
}


def cpu_baseline(cfg, stream, n_pre, steps, builds=("O3_native", "Og_fastmath", "O2_parity")):
    """SURVEY.md 8(d) / BASELINE.md section 3: the CPU path -- the same host orchestration over the CPU oracle backend -- on the
    same stream and the same frames as `value`, one thread pinned to one core, in the builds the survey names: `-O3
    -march=native` (strong baseline; the >= 200x target is quoted against it), the reference's own flags (`-Og ...
    -ffast-math -mtune=native` for the oracle; the orchestration, whose results must not change, without -ffast-math), and the
    parity build.  Each build is compiled on THIS host and timed in a child process."""
    import pickle
    import subprocess
    import tempfile

    import oracle
    from rd_vio_amd import build as rbuild

    out = {}
    tables = None
    with tempfile.TemporaryDirectory() as td:
        for name in builds:
            flags, variant = CPU_BUILDS[name]
            try:
                lib_path = rbuild.build_pipeline_variant(name, flags) if flags else rbuild.PIPE_LIB
                if variant:
                    oracle.build_backend(variant)
                pk = os.path.join(td, f"{name}.pkl")
                with open(pk, "wb") as fh:
                    pickle.dump((cfg, stream, n_pre, steps, variant, lib_path), fh)
                npz = os.path.join(td, f"{name}.npz")
                cmd = [sys.executable, os.path.abspath(__file__), "--cpu-path-worker", pk, "--cpu-path-out", npz]
                try:
                    core = sorted(os.sched_getaffinity(0))[-1]
                    cmd = ["taskset", "-c", str(core)] + cmd
                except (AttributeError, OSError):
                    pass
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
                if res.returncode != 0:
                    out[name] = {"error": (res.stderr or res.stdout)[-300:]}
                    continue
                rep = json.loads(res.stdout.strip().splitlines()[-1])
                oflags = " ".join(oracle.VARIANTS[variant]) if variant else "-O2 -ffp-contract=off -fno-fast-math"
                out[name] = {"value": round(rep["frames"] / rep["seconds"], 3), "unit": "frames/s", "frames": rep["frames"], "seconds": round(rep["seconds"], 2),
                             "flags": {"oracle": "gcc " + oflags, "orchestration": "g++ " + (" ".join(flags) if flags else "-O2") + " -ffp-contract=off"},
                             "backend_ms_per_frame": rep["backend_ms_per_frame"]}
                if name == "O2_parity":
                    tables = dict(np.load(npz))
            except Exception as exc:  # noqa: BLE001
                out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    head = out.get("O3_native") if "value" in out.get("O3_native", {}) else out.get("O2_parity", {})
    rep = dict(value=head.get("value"), unit="frames/s", cores=1, kind="port",
               sample=f"the {steps} timed frames of the same stream (after the same {n_pre} bootstrap + warm-up frames) through the same host orchestration "
                      f"over the CPU oracle backend (oracle/*.c, oracle/backend/), pipelined schedule on one thread pinned to one core, gcc -O3 "
                      f"-march=native for oracle and orchestration, {head.get('seconds')} s; other builds under `builds`",
               builds=out)
    return rep, tables


def compare_paths(gpu_run, cpu_tables, stream, cfg):
    """SURVEY.md 8d metrics (2) and (3): per-frame feature index sets / pixel positions and trajectories of the two paths"""
    from rd_vio_amd import pipeline_run as pr
    from rd_vio_amd import synth

    sg = gpu_run.table("window")
    kg = gpu_run.keypoints()
    sc = cpu_tables["window"]
    off = np.concatenate([[0], np.cumsum(cpu_tables["kp_n"])])
    kc = [(cpu_tables["kp_ids"][off[i]:off[i + 1]], cpu_tables["kp_xy"][off[i]:off[i + 1]]) for i in range(len(cpu_tables["kp_n"]))]
    n = min(len(kg), len(kc))
    same_idx = len(kg) == len(kc) and all(np.array_equal(kg[i][0], kc[i][0]) for i in range(n))
    same_px = same_idx and all(np.array_equal(kg[i][1], kc[i][1]) for i in range(n))
    m = min(len(sg), len(sc))
    both = ~np.isnan(sg[:m, 0]) & ~np.isnan(sc[:m, 0])
    out = {"frames_compared": int(n), "feature_indices_identical": bool(same_idx), "pixel_positions_identical": bool(same_px)}
    if both.sum() >= 3:
        a, b = sg[:m][both, 5:8], sc[:m][both, 5:8]
        out["ate_rmse_gpu_vs_cpu_path_mm"] = float(f"{1e3 * ate_rmse(a, b):.3g}")
        out["max_position_difference_mm"] = float(f"{1e3 * float(np.abs(a - b).max()):.3g}")
        p_gt = np.array([synth.traj_pose(t)[1] for t in sg[:m][both, 0]])
        out["ate_rmse_vs_ground_truth_m"] = {"gpu_path": round(ate_rmse(a, p_gt), 4), "cpu_path": round(ate_rmse(b, p_gt), 4),
                                             "note": "after rigid alignment (the initializer fixes its own world frame)"}
    return out


def ate_rmse(p_est, p_ref):
    """position RMSE after the best rigid (Umeyama, no scale) alignment of p_est onto p_ref"""
    a, b = np.asarray(p_est), np.asarray(p_ref)
    ma, mb = a.mean(0), b.mean(0)
    H = (a - ma).T @ (b - mb)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return float(np.sqrt(np.mean(np.sum(((a - ma) @ R.T + mb - b) ** 2, axis=1))))


def kernel_loop_leg(cfg, torch, dev, device_index, steps, warmup, serial, stream):
    """The resident kernel loop (rounds 1-2's `value`): one camera frame of the hot path on a fixed synthetic problem set with
    every input in HBM -- preprocess -> LK -> detect -> preintegration -> localize_newframe solve -> refine_window solve ->
    marginalisation, one host wait per frame, no host orchestration.  Per-stage times from HIP events on the launch streams."""
    import rd_vio_amd

    seq = Sequence(cfg, torch, dev, device_index, overlap=not serial, first_stream=stream)
    ctx, wl = seq.ctx, seq.wl
    for k in range(warmup):
        seq.step(k)
    N_EV, EV_PAIRS, names = Sequence.N_EV, Sequence.EV_PAIRS, Sequence.STAGES
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(N_EV)] for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        seq.step(warmup + k, evs[k])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stage_ms = np.zeros(len(names))
    for e in evs:
        for i, (a, b) in enumerate(EV_PAIRS):
            stage_ms[i] += e[a].elapsed_time(e[b])
    stage_ms /= steps
    _, _, sm_win = ctx.ba_fetch(0)
    _, _, sm_loc = ctx.ba_fetch(1)

    def hbm_row(kernel, nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return dict(kernel=kernel, bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 6),
                    algorithmic_bytes=int(nbytes), avg_stage_us=round(ms * 1e3, 2))

    P0 = wl["L"].w[0] * wl["L"].h[0]
    image_roof = [hbm_row("clahe_lut_kernel+pyr_level_kernel", preprocess_algorithmic_bytes(wl["L"]), stage_ms[0]),
                  hbm_row("lk_track_kernel", lk_algorithmic_bytes(cfg["features"]), stage_ms[1]),
                  hbm_row("harris_kernel+harris_candidates_kernel+gftt_select_kernel+poisson_filter_kernel (selection on the device)", 9 * P0, stage_ms[2])]
    pmc, pmc_note = pmc_traffic_bytes() if cfg is CONFIGS["euroc_v101"] else ({}, "PMC passes are taken on the default config only")
    if pmc:
        image_roof[0]["traffic"] = pmc.get("clahe_lut_kernel", 0) + pmc.get("pyr_level_kernel<true>", 0) + 3 * pmc.get("pyr_level_kernel<false>", 0)
        image_roof[1]["traffic"] = pmc.get("lk_track_kernel")
        image_roof[2]["traffic"] = pmc.get("harris_kernel", 0) + pmc.get("harris_candidates_kernel", 0)
    out = {"fps": round(steps / el, 2), "ms_per_step": round(1e3 * el / steps, 4), "steps": steps, "warmup": warmup,
           "schedule": "serial" if serial else "frontend lane overlapped with solver and marginalisation lanes, one host wait per frame",
           "stages_ms": {n: round(float(v), 4) for n, v in zip(names, stage_ms)},
           "ba_window": {"factors": int(len(wl["window_pb"]["tgt"])), "iterations": int(sm_win.iterations), "successful_steps": int(sm_win.successful_steps)},
           "ba_localize": {"factors": int(len(wl["localize_pb"]["tgt"])), "iterations": int(sm_loc.iterations)},
           "note": "fixed problem set: every frame is a keyframe whose window solve runs to the iteration limit -- the worst case of the cadence; "
                   "no host orchestration, no uploads",
           "image_kernels_roofline": image_roof}
    ctx.close()
    return out, pmc, pmc_note


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    # (the process reaches its steady rate ~150 frames after the first frame -- clocks, allocator, caches; measured per 100 frames: 1028,
    # 1122, 1080, 1334 ... against 1200, 1097, 1097, 1343 ... for a second pipeline in the same process -- so the default warm-up is
    # 100 frames behind the ~36 of the bootstrap)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="euroc_v101", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-loop", action="store_true", help="skip the resident kernel loop leg")
    ap.add_argument("--no-process-warmup", action="store_true", help="skip the throw-away pipeline that warms the process up before the timed one")
    ap.add_argument("--no-variants", action="store_true", help="skip the schedule / bootstrap variants of the pipeline leg")
    ap.add_argument("--threading", type=int, default=2, choices=(0, 1, 2), help="schedule of the timed pipeline (default: the product's, 2)")
    ap.add_argument("--sequences", type=int, default=0,
                    help="extra leg (1 GPU, rank 0): this many independent sequences sharing the GPU through the resident kernel loop (0 = skip)")
    ap.add_argument("--sequence-wait", type=int, default=-1, choices=(-1, 0, 1))
    ap.add_argument("--sequence-lanes", type=int, default=1, choices=(1, 3))
    ap.add_argument("--serial", action="store_true", help="kernel loop leg: one stream, stages back to back")
    # internal modes
    ap.add_argument("--cpu-path-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-path-out", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--stub-step-ms", type=float, default=None,
                    help="test mode: no GPU, a step is a sleep of this many ms (exercises the replica launcher and the timing protocol on CPU)")
    return ap.parse_args(argv)


def launch_replicas(args, argv):
    """`bench.py --gpus N` outside torchrun: N independent replicas, one child process and one GPU each (SURVEY.md 8e:
    replicas only, no data-path collective).  This parent never imports torch and never touches HIP; each child sees
    exactly one device (HIP_VISIBLE_DEVICES=i), rendezvous is gloo on 127.0.0.1, rank 0's JSON line (the aggregate over
    all replicas, with the per-replica rates) is relayed to stdout."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RDVIO_BENCH_DEVICE="0")
        if args.stub_step_ms is None:
            env["HIP_VISIBLE_DEVICES"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(rcs):
        sys.stderr.write(f"bench.py: replica exit codes {rcs}\n")
        sys.stdout.write(out0 or "")
        raise SystemExit(1)
    line = [l for l in (out0 or "").splitlines() if l.startswith("{")][-1]
    print(line)


def run_stub(args):
    """test mode (no GPU): the launcher, the gloo rendezvous, the barrier-bracketed timed region and the aggregate"""
    from rd_vio_amd import replica

    rank, local_rank, world, dist = replica.init_distributed("gloo")
    for _ in range(args.warmup):
        time.sleep(args.stub_step_ms * 1e-3)
    elapsed, per_rank = replica.timed_region(lambda k: time.sleep(args.stub_step_ms * 1e-3 * (1 + rank)), args.steps, sync=lambda: None,
                                             dist=dist, first_index=args.warmup, per_rank=True)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(world * args.steps / elapsed, 3), "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                          "per_replica_steps_per_s": [round(args.steps / e, 3) for e in per_rank]}))
    if dist is not None:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.cpu_path_worker:
        return cpu_path_worker(args)
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # torchrun (the driver's N > 1 launch) or launch_replicas
    if args.gpus > 1 and not under_launcher:
        return launch_replicas(args, argv)
    if args.stub_step_ms is not None:
        return run_stub(args)
    cfg = CONFIGS[args.config]
    if args.sequences > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(max(4, 2 * args.sequences * args.sequence_lanes), 32)))

    import ctypes

    # the synthetic stream is rendered by a pool of forked workers: before anything in this process touches the GPU
    steps, warmup = args.steps, args.warmup
    n_total = BOOT_FRAMES + 24 + warmup + steps
    stream = make_stream(cfg, n_total)

    import torch

    from rd_vio_amd import replica

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # torchrun: one process per GPU, device = LOCAL_RANK; launch_replicas: each child sees one device (index 0)
    local_rank = int(os.environ.get("RDVIO_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one process per GPU, replicas only: the only distributed traffic is the barrier and the max-over-ranks of a timer,
    # which gloo carries on the host -- no RCCL communicator is created for a path that has no collective
    rank, _lr, world, dist = replica.init_distributed("gloo")

    import rd_vio_amd
    from rd_vio_amd import build as rbuild
    from rd_vio_amd import pipeline_run as pr

    rbuild.build()
    lib = pr.load_pipeline_lib()
    def hip_context():
        return rd_vio_amd.Context(max_width=cfg["width"], max_height=cfg["height"], max_features=max(1024, 4 * cfg["features"]),
                                  max_window=cfg["window"] + 8, max_factors=40000, device=local_rank)

    def hip_run(threading, init_states=None, kp_capacity=0, **extra):
        ctx = hip_context()
        pcfg, applied = pipeline_config(lib, cfg, threading, **extra)
        run = PipelineRun(lib, lambda out: lib.rdvio_pipeline_create_hip(out, ctypes.byref(pcfg), ctx._h), stream, init_states, kp_capacity)
        return ctx, run, applied

    # ---- the timed pipeline
    supplied = cfg.get("bootstrap") == "groundtruth"
    if not args.no_process_warmup:
        # Process warm-up (untimed, like W): the first pipeline of a process runs 5-10 % slower than any later one (measured on the same
        # stream in the same process: first 1124, later 1242 frames/s; with this warm-up the first one reaches 1200 -- clocks, allocator,
        # page tables); the driver times one process per GPU from its start, so a throw-away pipeline replays 290 frames first.
        c0_, r0_, _ = hip_run(args.threading, init_states=stream["gt"] if supplied else None)
        try:
            bootstrap_and_warm_up(r0_, min(250, n_total - 2 * BOOT_FRAMES), n_total)
        finally:
            r0_.close()
            c0_.close()
    ctx, run, applied = hip_run(args.threading, init_states=stream["gt"] if supplied else None, kp_capacity=2048)
    n_pre = bootstrap_and_warm_up(run, warmup, n_total - steps)
    c0 = run.counters()
    # Every solver launch of the timed region carries HIP start / stop events (attached to the launch itself).  A timed launch costs
    # the lane a few microseconds (its stop event keeps the chained next solve from following it directly): measured on `value`,
    # between nothing and 6 % depending on the box, inside the box-to-box spread.  RDVIO_BENCH_KT_STRIDE=k times the launches of one
    # frame in k (picked by a hash of the frame index: a fixed stride aliases with the keyframe cadence), 0 none -- for that A/B.
    KT_STRIDE = int(os.environ.get("RDVIO_BENCH_KT_STRIDE", "1"))
    ctx._check(ctx._lib.rdvio_hip_ctx_set_kernel_timing(ctx._h, KT_STRIDE))
    timed = {}

    def block(_first, k):
        timed["out"] = run.segment(k)   # exactly k frames: replay + drain of the estimator's last step

    elapsed, per_rank = replica.timed_region(None, steps, sync=torch.cuda.synchronize, dist=dist, first_index=n_pre, per_rank=True, block=block)
    kt = np.zeros(4)
    ctx._check(ctx._lib.rdvio_hip_ctx_get_kernel_timing(ctx._h, kt.ctypes.data_as(ctypes.c_void_p)))
    c1 = run.counters()
    assert timed["out"]["frames_processed"] == steps, timed["out"]["frames_processed"]

    if rank == 0:
        value = world * steps / elapsed
        names = pr.COUNTER_NAMES
        dc = c1 - c0
        backend_ms = {n: round(float(dc[11 + 2 * k]) / 1e3 / steps, 4) for k, n in enumerate(names)}
        launches, k_ms, k_flops = float(kt[0]), float(kt[1]), float(kt[2])
        tfl = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        out = {
            "metric": "VIO frames/sec per GPU (product pipeline: LK tracker + RD path + sliding-window BA), synthetic EuRoC-shaped stream",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int64 (image, LK) + f64 (estimation)", "data": "synthetic",
            "config": {"workload": cfg["name"], "features": cfg["features"], "window": cfg["window"], "replicas": world,
                       "settings": "configs/baseline_setting.yaml (the reference's configs/setting.yaml values) with only sliding_window.size and "
                                   "feature_tracker.max_keypoint_detection overridden; configs/synthetic_euroc_sensor.yaml",
                       "settings_applied": {k: (v if not isinstance(v, (list, tuple, np.ndarray)) else [float(x) for x in v]) for k, v in applied.items()},
                       "bootstrap": "supplied keyframe states (ground truth of the synthetic stream) instead of the SfM / IMU-alignment stages" if supplied
                                    else "full initializer (SfM + IMU alignment), no supplied states",
                       "schedule": {0: "inline (the reference's THREADING=OFF)", 1: "pipelined tracker / frontend schedule on one thread",
                                    2: "tracker on the caller's thread, frontend's step on a worker thread (frontend lane || solver lane), "
                                       "deterministic hand-over"}[args.threading],
                       "frames": {"bootstrap_and_warmup_untimed": int(n_pre), "timed": steps},
                       "timed_region": {"window_solves": int(dc[1]), "marginalizations": int(dc[3]), "localizations": int(dc[4]), "subwindow_solves": int(dc[5]),
                                        "solver_iterations": int(dc[10]), "imu_parsac_judgements": int(dc[27]), "tracks_marked_dynamic": int(dc[28]),
                                        "largest_solve": {"frames": int(c1[8]), "factors": int(c1[9])}}},
            "backend_ms_per_frame": backend_ms,
            "per_replica_fps": [round(steps / e, 2) for e in per_rank],
            # dominant kernel: ba_solve_kernel (FP64, one persistent workgroup per solve: a dependent-latency chain, neither roofline
            # binds it -- DESIGN.md section 4).  achieved = SURVEY 8(d) algorithmic flops of the launches of the timed region / their
            # HIP-event durations (events on the solver lane, the stream the kernel is launched on)
            "roofline": dict(kernel="ba_solve_kernel", bound="mfma", achieved=round(tfl, 6), peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=round(tfl / FP64_PEAK_TFLOPS, 8), traffic=None,
                             launches_timed=int(launches), timed_every=KT_STRIDE,
                             launches=int(dc[1] + dc[4] + dc[5]), launches_per_frame=round(float(dc[1] + dc[4] + dc[5]) / steps, 3),
                             avg_launch_us=round(1e3 * k_ms / max(launches, 1), 2),
                             algorithmic_flops_per_launch=int(k_flops / max(launches, 1)), solver_iterations_timed=int(kt[3]),
                             kernel_ms_per_frame=round((k_ms / max(launches, 1)) * float(dc[1] + dc[4] + dc[5]) / steps, 4),
                             note="measured live over the timed region with HIP start / stop events attached to every launch on the solver lane "
                                  "(localize_newframe + refine_window or refine_subwindow, as the stream produced them); single-workgroup "
                                  "latency-bound trust-region loop (phase table: DESIGN.md section 4)"),
        }

        # HBM-side bytes per launch of the dominant kernel: the committed PMC passes of this same command (pmc_traffic_bytes)
        pmc_main, pmc_main_note = pmc_traffic_bytes() if cfg is CONFIGS["euroc_v101"] else ({}, "PMC passes are taken on the default config only")
        out["roofline"]["traffic"] = pmc_main.get("ba_solve_kernel")
        out["roofline"]["traffic_unit"] = pmc_main_note

        def leg(name, fn):   # the legs beside `value`: a failure in one of them is recorded in its own field and never costs the line
            try:
                out[name] = fn()
            except Exception as exc:  # noqa: BLE001
                out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

        cpu_tables = None
        if world == 1 and not args.no_cpu_baseline:
            def _cpu():
                nonlocal cpu_tables
                rep, cpu_tables = cpu_baseline(cfg, stream, n_pre, steps)
                return rep
            leg("cpu_baseline", _cpu)
            if isinstance(out.get("cpu_baseline"), dict) and out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 2)
            if cpu_tables is not None:
                leg("end_to_end", lambda: compare_paths(run, cpu_tables, stream, cfg))
    run.close()
    ctx.close()

    if rank == 0 and world == 1:
        if not args.no_variants:
            def _variants():
                rep = {}
                own = stream["gt"] if supplied else None
                other = ("full_initializer_bootstrap", args.threading, None, {}) if supplied else ("groundtruth_bootstrap", args.threading, stream["gt"], {})
                # (`value` is measured with the live kernel timing on -- two HIP events around every solver launch, read at the fetch --
                # which the roofline needs and a deployment does not: the first variant is the same run without it)
                for label, thr, init, extra in (("without_live_kernel_timing", args.threading, own, {}), ("inline_schedule", 0, own, {}), other,
                                                ("tracker_gates_on_the_device", args.threading, own, {"tracker_gates_on_backend": 1})):
                    c2, r2, _ = hip_run(thr, init_states=init, **extra)
                    try:
                        n2 = bootstrap_and_warm_up(r2, warmup, n_total - steps)
                        t2 = r2.segment(min(steps, n_total - n2))
                        rep[label] = {"fps": round(t2["frames_processed"] / t2["elapsed_s"], 2), "frames": int(t2["frames_processed"]),
                                      "bootstrap_and_warmup_frames": int(n2)}
                    finally:
                        r2.close()
                        c2.close()
                return rep
            leg("pipeline_variants", _variants)
        if not args.no_kernel_loop:
            def _kl():
                s2 = torch.cuda.Stream(device=dev)
                torch.cuda.set_stream(s2)
                rep, pmc, pmc_note = kernel_loop_leg(cfg, torch, dev, local_rank, min(steps, 100), 10, args.serial, s2)
                return rep
            leg("kernel_loop", _kl)
        if args.sequences > 1:
            def _ms():
                s3 = torch.cuda.Stream(device=dev)
                torch.cuda.set_stream(s3)
                first = Sequence(cfg, torch, dev, local_rank, overlap=True, first_stream=s3)
                try:
                    return multi_sequence(cfg, torch, dev, local_rank, first.wl, args.sequences, steps=min(steps, 100), warmup=10,
                                          overlap=args.sequence_lanes == 3, wait_mode=args.sequence_wait)
                finally:
                    first.ctx.close()
            leg("multi_sequence", _ms)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
