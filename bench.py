#!/usr/bin/env python3
"""bench.py -- frames/sec of rd_vio's per-frame hot path on MI355X (BASELINE.json metric).

A "step" is one camera frame of the hot path on a synthetic EuRoC-shaped stream (there is no EuRoC data on
the box): CLAHE + 4-level pyramid + Scharr (A1) -> fused forward/backward pyramidal LK (A2) -> GFTT-Harris
detection (A3) -> IMU preintegration of the frame segment and the W keyframe segments (A7) ->
localize_newframe solve, refine_window solve (A8-A14) and marginalisation of the oldest frame (A13), with every
input resident in HBM before the timed
region.  Multi-GPU = one independent replica (one stream) per GPU, no collective on the data path
(SURVEY.md 8e); torch.distributed is used only for the barrier / max-over-ranks timing.

Prints ONE JSON line (see the round contract): metric/value/unit + `roofline` (dominant kernel, measured with
HIP events on the launch stream) + `cpu_baseline` (the CPU oracle timed on this box, rank 0, N=1 only).
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
    "euroc_mh03_rd": dict(width=752, height=480, features=300, window=10, landmarks=300, iters=30, parsac=True,
                          name="EuRoC MH_03_medium-shaped synthetic stream 752x480, 300 features, window 10, RD path"),
    # configs[4]: roofline run
    "synthetic_720p": dict(width=1280, height=720, features=1000, window=16, landmarks=1000, iters=30,
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


PMC_TAG = "r02"


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


CPU_STAGES = ["preprocess", "lk_track", "detect", "preintegrate", "ba_localize", "ba_window", "marginalize"]


def cpu_baseline_loop(cfg, wl, budget_s, max_frames):
    """The CPU oracle (single thread) on the same per-frame workload; bounded sample; per-stage wall time."""
    import oracle
    from rd_vio_amd import synth

    oracle.build()
    frames = wl["frames_host"]
    kp = wl["kp_host"]
    pyr = oracle.preprocess(frames[0])
    n = 0
    stage = np.zeros(len(CPU_STAGES))
    t0 = time.perf_counter()
    while True:
        img = frames[(n + 1) % len(frames)]
        a = time.perf_counter()
        nxt = oracle.preprocess(img)
        b = time.perf_counter()
        oracle.track_keypoints(pyr[0], (pyr[1], pyr[2]), (nxt[1], nxt[2]), kp)
        c = time.perf_counter()
        lvl0 = np.ascontiguousarray(oracle.level_view(nxt[0], nxt[1], 0))
        oracle.detect_keypoints(lvl0, np.zeros((0, 2)), cfg["features"], 10.0)
        d = time.perf_counter()
        for i, s in enumerate(wl["imu_segs"]):
            p = wl["imu_par"][i]
            oracle.preintegrate(s, p[0], p[1:4], p[4:7], synth.EUROC_NOISE, jac=i > 0, cov=i > 0)
        e = time.perf_counter()
        oracle.ba_solve(wl["localize_pb"], cfg["iters"])
        f = time.perf_counter()
        oracle.ba_solve(wl["window_pb"], cfg["iters"])
        g = time.perf_counter()
        oracle.marginalize(*wl["marg_args"])
        h = time.perf_counter()
        stage += np.diff([a, b, c, d, e, f, g, h])
        pyr = nxt
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= max_frames:
            break
    return n, el, stage / n * 1e3


def cpu_baseline_worker(args):
    """child process of cpu_baseline(): loads the oracle build named by RDVIO_ORACLE_LIB (set by the parent), runs the
    bounded loop on the pickled workload, prints one JSON line.  Never touches the GPU."""
    import pickle

    with open(args.cpu_baseline_worker, "rb") as fh:
        cfg, wl = pickle.load(fh)
    n, el, stage = cpu_baseline_loop(cfg, wl, args.cpu_budget, 400)
    print(json.dumps({"frames": n, "seconds": el, "stage_ms": [float(x) for x in stage]}))


def cpu_baseline(cfg, wl, budget_s=10.0):
    """SURVEY.md 8(d) / BASELINE.md section 3: the CPU restatement of the same per-frame work on one pinned host core, in the
    two builds the survey names -- `-O3 -march=native` (strong baseline; the >= 200x target is quoted against it) and the
    reference's own flags `-Og -msse -msse2 -msse3 -ffast-math -mtune=native` (/root/reference/CMakeLists.txt:10,17-18) --
    each compiled on THIS host and timed in a child process (one thread, pinned to one core), with per-stage milliseconds.
    The parity build (-O2, no contraction, no fast-math) is timed too, for continuity with round 1."""
    import pickle
    import subprocess
    import tempfile

    import oracle

    keep = ("frames_host", "kp_host", "imu_segs", "imu_par", "localize_pb", "window_pb", "marg_args")
    with tempfile.NamedTemporaryFile(suffix=".pkl", delete=False) as fh:
        pickle.dump((cfg, {k: wl[k] for k in keep}), fh)
        path = fh.name
    builds = {}
    try:
        for name in ("O3_native", "Og_fastmath", "O2_parity"):
            env = dict(os.environ)
            if name == "O2_parity":
                env.pop("RDVIO_ORACLE_LIB", None)
                flags = "-O2 -ffp-contract=off -fno-fast-math"
            else:
                env["RDVIO_ORACLE_LIB"] = oracle.build_variant(name)
                flags = " ".join(oracle.VARIANTS[name])
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", path, "--cpu-budget", str(budget_s)]
            try:
                core = sorted(os.sched_getaffinity(0))[-1]
                cmd = ["taskset", "-c", str(core)] + cmd
            except (AttributeError, OSError):
                pass
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=20 * budget_s + 120)
            if out.returncode != 0:
                builds[name] = {"error": (out.stderr or out.stdout)[-300:], "flags": flags}
                continue
            rep = json.loads(out.stdout.strip().splitlines()[-1])
            builds[name] = {"value": round(rep["frames"] / rep["seconds"], 3), "unit": "frames/s", "flags": "gcc " + flags,
                            "frames": rep["frames"], "seconds": round(rep["seconds"], 2),
                            "stage_ms": {k: round(v, 4) for k, v in zip(CPU_STAGES, rep["stage_ms"])}}
    finally:
        os.unlink(path)
    head = builds.get("O3_native", {})
    if "value" not in head:
        head = builds.get("O2_parity", {})
    return dict(value=head.get("value"), unit="frames/s", cores=1, kind="port",
                sample=f"{head.get('frames')} frames of the same synthetic per-frame work through the CPU oracle (oracle/*.c), gcc -O3 "
                       f"-march=native, 1 thread pinned to one core, {head.get('seconds')} s; other builds under `builds`",
                stage_ms=head.get("stage_ms"), builds=builds)


def end_to_end(cfg, ctx, n_frames, with_cpu_path):
    """SURVEY.md 8d metric (1)-(3) on a geometrically consistent synthetic stream: the product pipeline
    (librdvio_pipeline.so: feature tracker + sliding-window tracker over the HIP backend) fed like test_euroc feeds
    rdvio::Odometry.  This rate is PCIe-inclusive (host images in, host-built BA graphs in, states out every frame), so
    it is reported beside `value`, never as `value`.  With `with_cpu_path` the same orchestration runs over the CPU
    oracle backend (cpu_baseline leg) for the trajectory / feature-index comparison."""
    import ctypes

    from rd_vio_amd import pipeline_run as pr
    from rd_vio_amd import synth

    w, h = cfg["width"], cfg["height"]
    K = synth.EUROC_K.copy()
    if (w, h) != (752, 480):
        K = np.array([[900.0, 0, w / 2.0], [0, 900.0, h / 2.0], [0, 0, 1.0]])
    frames, ts, imu, gt = synth.make_stream(n_frames, w, h, K, mover=bool(cfg.get("parsac")))
    lib = pr.load_pipeline_lib()
    pcfg = pr.default_config(lib, K, w, h, synth.EUROC_EXTR, synth.EUROC_NOISE, sliding_window_size=cfg["window"],
                             feature_tracker_max_keypoint_detection=cfg["features"], feature_tracker_min_keypoint_distance=10.0,
                             solver_iteration_limit=cfg["iters"], feature_tracker_max_frames=20,
                             sliding_window_force_keyframe_landmarks=50, sliding_window_subframe_size=3,
                             rotation_misalignment_threshold=0.02, parsac_flag=1 if cfg.get("parsac") else 0, parsac_keyframe_check_size=1)
    gt_c = np.ascontiguousarray(gt)

    def run(handle):
        states, kps, stamps = [], [], []
        ids = np.zeros(4096, dtype=np.int64)
        xy = np.zeros((4096, 2))
        st16 = np.zeros(16)
        tt = ctypes.c_double(0)
        t_start = time.perf_counter()

        def snap(_n):
            n = lib.rdvio_pipeline_last_frame_keypoints(handle, ids.ctypes.data_as(ctypes.c_void_p), xy.ctypes.data_as(ctypes.c_void_p), 4096)
            kps.append((ids[:n].copy(), xy[:n].copy()))
            ok = lib.rdvio_pipeline_window_state(handle, ctypes.byref(tt), st16.ctypes.data_as(ctypes.c_void_p))
            states.append(np.concatenate([[tt.value], st16]) if ok else np.full(17, np.nan))
            stamps.append(time.perf_counter() - t_start)

        assert lib.rdvio_pipeline_set_init_states(handle, len(gt_c), gt_c.ctypes.data_as(ctypes.c_void_p)) == 0
        spent = pr.feed_stream(lib, handle, frames, ts, imu, per_frame=snap)
        cnt = np.zeros(29, dtype=np.int64)
        lib.rdvio_pipeline_counters(handle, cnt.ctypes.data_as(ctypes.c_void_p))
        lib.rdvio_pipeline_destroy(handle)
        return np.array(states), kps, np.array(stamps), spent, cnt

    sg, kg, stamps, spent, cnt = run(pr.create_hip_pipeline(lib, ctx, pcfg))
    tracking = ~np.isnan(sg[:, 0])
    out = {"frames": int(cnt[0]), "frames_tracking": int(tracking.sum()), "fps": round(float(cnt[0] / spent), 2),
           "window_solves": int(cnt[1]), "marginalizations": int(cnt[3]), "localizations": int(cnt[4]), "subwindow_solves": int(cnt[5]),
           "largest_solve": {"frames": int(cnt[8]), "factors": int(cnt[9])}, "solver_iterations": int(cnt[10]),
           "rd_path": {"imu_parsac_judgements": int(cnt[27]), "tracks_marked_dynamic": int(cnt[28])},
           "backend_ms_per_frame": {name: round(float(cnt[11 + 2 * k]) / 1e3 / max(int(cnt[0]), 1), 4) for k, name in enumerate(
               ("preprocess", "detect", "track", "preintegrate", "ba_solve", "marginalize", "image_create"))},
           "backend_calls": {name: int(cnt[12 + 2 * k]) for k, name in enumerate(
               ("preprocess", "detect", "track", "preintegrate", "ba_solve", "marginalize", "image_create"))},
           "ms_per_frame": round(1e3 * spent / max(int(cnt[0]), 1), 4),
           "note": "product pipeline over the HIP backend, host buffers in/out every frame (PCIe-inclusive)"}
    if tracking.sum() >= 2:
        i0 = int(np.argmax(tracking))
        out["fps_tracking_phase"] = round(float((len(stamps) - 1 - i0) / (stamps[-1] - stamps[i0])), 2)
        p_gt = np.array([synth.traj_pose(t)[1] for t in sg[tracking, 0]])
        out["position_error_vs_ground_truth_m"] = {"max": round(float(np.linalg.norm(sg[tracking, 5:8] - p_gt, axis=1).max()), 4),
                                                   "final": round(float(np.linalg.norm(sg[tracking, 5:8][-1] - p_gt[-1])), 4)}
    if with_cpu_path:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pipeline_util as pu

        shim = pu.build_oracle_backend()
        hc = ctypes.c_void_p()
        assert pu.oracle_pipeline_factory(lib, shim, pcfg)(ctypes.byref(hc)) == 0
        sc, kc, stamps_c, spent_c, cnt_c = run(hc)
        same_idx = len(kc) == len(kg) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(kg, kc))
        both = tracking & ~np.isnan(sc[:, 0]) if len(sc) == len(sg) else np.zeros(0, dtype=bool)
        out["cpu_path"] = {"fps": round(float(cnt_c[0] / spent_c), 2),
                           "feature_indices_identical": bool(same_idx),
                           "ate_rmse_gpu_vs_cpu_path_mm": float(f"{1e3 * pu.ate_rmse(sg[both, 5:8], sc[both, 5:8]):.3g}") if both.sum() >= 3 else None,
                           "max_position_difference_mm": float(f"{1e3 * float(np.abs(sg[both, 5:8] - sc[both, 5:8]).max()):.3g}") if both.sum() else None,
                           "kind": "the same orchestration over the CPU oracle backend (tests/cpp/oracle_backend.c), 1 thread"}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="euroc_v101", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--end-to-end-frames", type=int, default=100, help="frames of the pipeline run (0 = skip)")
    ap.add_argument("--sequences", type=int, default=16,
                    help="extra leg (1 GPU, rank 0): this many independent sequences sharing the GPU, aggregate frames/s reported beside `value` (0 = skip)")
    ap.add_argument("--sequence-wait", type=int, default=-1, choices=(-1, 0, 1), help="host waits of the multi-sequence leg: 0 spin, 1 block, -1 auto")
    ap.add_argument("--sequence-lanes", type=int, default=1, choices=(1, 3),
                    help="streams per sequence in the multi-sequence leg: 1 = stages back to back (the device overlaps ACROSS sequences), 3 = the lanes of `value`")
    ap.add_argument("--serial", action="store_true", help="one stream, stages back to back (the round-1 step); default: frontend / estimator streams overlapped")
    # internal modes
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=10.0, help=argparse.SUPPRESS)
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
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # torchrun (the driver's N > 1 launch) or launch_replicas
    if args.gpus > 1 and not under_launcher:
        return launch_replicas(args, argv)
    if args.stub_step_ms is not None:
        return run_stub(args)
    cfg = CONFIGS[args.config]
    if args.sequences > 1:
        # the multi-sequence leg runs one HIP stream per sequence: the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
        # (default 4), and streams that share a queue run one after the other -- ask for one queue per sequence before HIP starts
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(max(4, 2 * args.sequences * args.sequence_lanes), 32)))

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

    from rd_vio_amd import build as rbuild
    import rd_vio_amd

    rbuild.build()
    # explicit (non-default) streams: the context enqueues on them and torch.cuda.Event records on them, so the HIP events
    # below bracket exactly the kernels of each stage.  Lanes (include/rdvio_hip.h): the frontend (image side +
    # preintegration), the solver and the marginalisation each get their own stream unless --serial.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    seq = Sequence(cfg, torch, dev, local_rank, overlap=not args.serial, first_stream=stream)
    ctx, wl, step, overlap = seq.ctx, seq.wl, seq.step, seq.overlap
    nfeat, iters = cfg["features"], cfg["iters"]
    stage_names, NS, EV_PAIRS, N_EV = Sequence.STAGES, len(Sequence.STAGES), Sequence.EV_PAIRS, Sequence.N_EV

    # slot 1 holds frame 0 (build_workload); warm up
    for k in range(args.warmup):
        step(k)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(N_EV)] for _ in range(args.steps)]
    elapsed, per_rank = replica.timed_region(lambda k: step(k, evs[k - args.warmup]), args.steps, sync=torch.cuda.synchronize,
                                             dist=dist, first_index=args.warmup, per_rank=True)
    stage_ms = np.zeros(NS)
    for e in evs:
        for i, (a, b) in enumerate(EV_PAIRS):
            stage_ms[i] += e[a].elapsed_time(e[b])
    stage_ms /= args.steps
    _, _, sm_win = ctx.ba_fetch(0)
    _, _, sm_loc = ctx.ba_fetch(1)

    if rank == 0:
        value = world * args.steps / elapsed
        stages = {n: round(float(v), 4) for n, v in zip(stage_names, stage_ms)}
        dom = int(np.argmax(stage_ms))

        def hbm_row(kernel, nbytes, ms):
            gbs = nbytes / (ms * 1e-3) / 1e9
            return dict(kernel=kernel, bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(gbs / HBM_PEAK_GBS, 6), algorithmic_bytes=int(nbytes), avg_stage_us=round(ms * 1e3, 2))

        # the dominant kernel is ba_solve_kernel (two launches per frame: localize_newframe and refine_window): FP64,
        # one workgroup, bounded by dependent-latency chains rather than by either roofline; it is priced against
        # the FP64 matrix peak because its dense pieces run on MFMA
        fl = (ba_algorithmic_flops(wl["window_pb"], sm_win.iterations, sm_win.successful_steps)
              + ba_algorithmic_flops(wl["localize_pb"], sm_loc.iterations, sm_loc.successful_steps))
        ba_ms = stage_ms[4] + stage_ms[5]
        tfl = fl / (ba_ms * 1e-3) / 1e12
        pmc, pmc_note = pmc_traffic_bytes() if args.config == "euroc_v101" else ({}, "PMC passes are taken on the default config only")
        roof = dict(kernel="ba_solve_kernel", bound="mfma", achieved=round(tfl, 5), peak=FP64_PEAK_TFLOPS,
                    unit="TFLOP/s", frac=round(tfl / FP64_PEAK_TFLOPS, 7), traffic=pmc.get("ba_solve_kernel"),
                    traffic_unit=pmc_note,
                    algorithmic_flops_per_launch=int(fl / 2), avg_launch_us=round(float(ba_ms) * 1e3 / 2, 2),
                    launches_per_frame=2, dominant_stage=stage_names[dom],
                    note="single-workgroup latency-bound trust-region loop; see DESIGN.md section 4 for the phase table")
        # image-side kernels against the HBM roofline (algorithmic bytes from SURVEY.md 8d)
        P0 = wl["L"].w[0] * wl["L"].h[0]
        image_roof = [hbm_row("clahe_lut_kernel+pyr_level_kernel", preprocess_algorithmic_bytes(wl["L"]), stage_ms[0]),
                      hbm_row("lk_track_kernel", lk_algorithmic_bytes(nfeat), stage_ms[1]),
                      hbm_row("harris_kernel+harris_candidates_kernel (+host selection)", 9 * P0, stage_ms[2])]
        if pmc:
            image_roof[0]["traffic"] = pmc.get("clahe_lut_kernel", 0) + pmc.get("pyr_level_kernel<true>", 0) + 3 * pmc.get("pyr_level_kernel<false>", 0)
            image_roof[1]["traffic"] = pmc.get("lk_track_kernel")
            image_roof[2]["traffic"] = pmc.get("harris_kernel", 0) + pmc.get("harris_candidates_kernel", 0)
        out = {
            "metric": "VIO frames/sec per GPU (hot path: LK tracker + sliding-window BA), synthetic EuRoC-shaped stream",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int64 (image, LK) + f64 (estimation)", "data": "synthetic",
            "config": {"workload": cfg["name"], "features": nfeat, "window": cfg["window"],
                       "solver_iteration_limit": iters, "replicas": world,
                       "schedule": ("frontend lane (image side + preintegration of frame k+1) overlapped with the solver lane (localisation + "
                                    "window solve of frame k) and the marginalisation lane; one host wait per frame on frontend + solver"
                                    if overlap else "serial: all stages back to back on one stream, one host wait per frame"),
                       "fixed_problem_note": "every frame is a keyframe whose window solve runs to the iteration limit (27 of 30 trial steps "
                                             "rejected through the reference's live bias-linearisation, DESIGN.md section 2) -- a reading of "
                                             "Ceres that no reference fixture pins",
                       "ba_window": {"factors": int(len(wl["window_pb"]["tgt"])), "iterations": int(sm_win.iterations),
                                     "successful_steps": int(sm_win.successful_steps)},
                       "ba_localize": {"factors": int(len(wl["localize_pb"]["tgt"])), "iterations": int(sm_loc.iterations)}},
            "stages_ms": stages,
            "per_replica_fps": [round(args.steps / e, 2) for e in per_rank],
            "roofline": roof,
            "image_kernels_roofline": image_roof,
        }
        # the legs beside `value`: a failure in one of them is recorded in its own field and never costs the line
        def leg(name, fn):
            try:
                out[name] = fn()
            except Exception as exc:  # noqa: BLE001
                out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

        if world == 1 and not args.no_cpu_baseline:
            leg("cpu_baseline", lambda: cpu_baseline(cfg, wl))
            if out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 2)
        if world == 1 and args.end_to_end_frames > 0:
            leg("end_to_end", lambda: end_to_end(cfg, ctx, args.end_to_end_frames, with_cpu_path=not args.no_cpu_baseline))
        if world == 1 and args.sequences > 1 and not args.serial:
            leg("multi_sequence", lambda: multi_sequence(cfg, torch, dev, local_rank, wl, args.sequences, steps=min(args.steps, 100), warmup=10,
                                                         overlap=args.sequence_lanes == 3, wait_mode=args.sequence_wait))
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
