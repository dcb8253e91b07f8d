"""Developer timing aid: phase stamps of a refine_subwindow-shaped solve (K free frames behind a fixed keyframe, every landmark
fixed, a chain of preintegration factors) -- GPU box, RDVIO_PROF build for the table."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, ".")
import rd_vio_amd
from rd_vio_amd import synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = rd_vio_amd.Context(max_window=16, max_factors=20000)
pre = lambda imu, t, bg, ba: ctx.preintegrate([imu], [t], [bg], [ba], synth.EUROC_NOISE)[0]
W = 8
pb = synth.make_window_problem(W + 1, 150, 648, preintegrate=pre)
sub = dict(pb)
sub["frame_fixed"] = np.ones(W + 1, dtype=np.uint8); sub["frame_fixed"][W + 1 - K:] = 0
sub["lm_fixed"] = np.ones(len(pb["inv_depth"]), dtype=np.uint8)
keep = pb["tgt"] >= W + 1 - K
for k in ("tgt", "ref", "lm", "tangent"):
    sub[k] = pb[k][keep]
sub["pre_i"], sub["pre_j"], sub["preint"] = pb["pre_i"][-K:], pb["pre_j"][-K:], pb["preint"][-K:]
for k in ("prior_frames", "lin", "S", "f"):
    sub.pop(k, None)
ctx.ba_upload(sub, 0)
for iters in (0, 1, 5, 30):
    ctx.ba_solve_resident(iters, 0); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.ba_solve_resident(iters, 0)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    _, _, sm = ctx.ba_fetch(0)
    print(f"subwindow K={K} F={len(sub['tgt'])} max_iter={iters:2d}: {dt*1e6:8.1f} us iterations={sm.iterations} successful={sm.successful_steps} term={sm.termination}")
if os.environ.get("RDVIO_PROF"):
    prof = np.zeros(72)
    ctx._lib.rdvio_hip_debug_ba_prof(ctx._h, 0, ctypes.c_void_p(prof.ctypes.data))
    names = ["setup", "eval_lin", "build_ne", "dogleg_prep", "schur", "cholesky", "tri_solve", "lm_y+norms",
             "step+model", "cand_eval", "misc", "gradmax", "ne:pairs", "ne:landm", "ne:preint", "ne:wait", "ne:phase2",
             "evL:factors(w0)", "evL:wait", "evL:whiten", "evC:factors(w0)", "evC:wait", "evC:whiten",
             "setup:copy/zero", "setup:stage S+ST", "setup:Lam gemm", "setup:eta0+mirror", "ne:H blocks", "candidate", "schur:lm_w+gemm", "evL:stage", "evC:stage"]
    for i, nm in enumerate(names):
        if prof[32 + i] > 0:
            print(f"      {nm:16s} total {prof[i] / 100:9.1f} us  calls {int(prof[32 + i]):3d}  avg {prof[i] / 100 / prof[32 + i]:8.2f} us")
