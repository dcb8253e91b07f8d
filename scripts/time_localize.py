"""Developer timing aid: phase stamps of the localize_newframe-shaped solve of bench.py (GPU box, RDVIO_PROF build)."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, ".")
import rd_vio_amd
from rd_vio_amd import synth

ctx = rd_vio_amd.Context(max_window=16, max_factors=20000)
pre = lambda imu, t, bg, ba: ctx.preintegrate([imu], [t], [bg], [ba], synth.EUROC_NOISE)[0]
W = 8
pb = synth.make_window_problem(W + 1, 150, 648, preintegrate=pre)
loc = dict(pb)
loc["frame_fixed"] = np.ones(W + 1, dtype=np.uint8); loc["frame_fixed"][W] = 0
loc["lm_fixed"] = np.ones(len(pb["inv_depth"]), dtype=np.uint8)
keep = pb["tgt"] == W
for k in ("tgt", "ref", "lm", "tangent"):
    loc[k] = pb[k][keep]
loc["pre_i"], loc["pre_j"], loc["preint"] = pb["pre_i"][-1:], pb["pre_j"][-1:], pb["preint"][-1:]
for k in ("prior_frames", "lin", "S", "f"):
    loc.pop(k, None)
ctx.ba_upload(loc, 0)
for iters in (0, 1, 5, 30):
    ctx.ba_solve_resident(iters, 0); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.ba_solve_resident(iters, 0)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    _, _, sm = ctx.ba_fetch(0)
    print(f"localize F={len(loc['tgt'])} max_iter={iters:2d}: {dt*1e6:8.1f} us iterations={sm.iterations} successful={sm.successful_steps} term={sm.termination}")
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
