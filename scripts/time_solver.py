"""Developer timing aid: wall time of the resident BA solve for problem variants (GPU box only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import rd_vio_amd
from rd_vio_amd import synth

ctx = rd_vio_amd.Context(max_window=16, max_factors=20000)
pre = lambda imu, t, bg, ba: ctx.preintegrate([imu], [t], [bg], [ba], synth.EUROC_NOISE)[0]


def timeit(pb, iters, reps=20):
    ctx.ba_upload(pb, 0)
    ctx.ba_solve_resident(iters, 0); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.ba_solve_resident(iters, 0)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    _, _, sm = ctx.ba_fetch(0)
    return dt * 1e6, sm.iterations, sm.successful_steps


W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for name, kw in (("full", {}), ("no_prior", dict(with_prior=False)), ("no_preint", dict(with_preint=False)),
                 ("vision_only", dict(with_prior=False, with_preint=False))):
    pb = synth.make_window_problem(W + 1, L, 648, preintegrate=pre, **kw)
    if not kw.get("with_preint", True):
        pb["frame_fixed"][:2] = 1
    for iters in (0, 1, 5, 30):
        us, it, ok = timeit(pb, iters)
        print(f"{name:12s} F={len(pb['tgt'])} max_iter={iters:2d}: {us:9.1f} us  iterations={it} successful={ok}")
        if os.environ.get("RDVIO_PROF") and iters == 30:
            import ctypes
            prof = np.zeros(72)
            ctx._lib.rdvio_hip_debug_ba_prof(ctx._h, 0, ctypes.c_void_p(prof.ctypes.data))
            names = ["setup", "eval_lin", "build_ne", "dogleg_prep", "schur", "cholesky", "tri_solve", "lm_y+norms",
                     "step+model", "cand_eval", "misc", "gradmax", "ne:pairs", "ne:landm", "ne:preint", "ne:wait", "ne:phase2",
                     "evL:factors(w0)", "evL:wait", "evL:whiten", "evC:factors(w0)", "evC:wait", "evC:whiten",
                     "setup:copy/zero", "setup:stage S+ST", "setup:Lam gemm", "setup:eta0+mirror", "ne:H blocks", "candidate", "schur:lm_w+gemm", "evL:stage", "evC:stage"]
            if os.environ.get("RDVIO_PROF_CHOL") and prof[64:69].any():
                print("      cholesky_lds, thread 0 (whole solve): panel %.1f us, tile (0,0) %.1f us, diagonal block %.1f us, at barriers %.1f us, diagonal inverses %.1f us" % tuple(prof[64:69] / 100))
            elif prof[64:68].any():
                print("      H blocks, wave 0: load issue %.1f us, load wait %.1f us, products + stores %.1f us" % tuple(prof[64:67] / 100))
            if prof[68:72].any():
                print("      shader clock during the group products: %.0f MHz (s_memtime ticks / s_memrealtime time)" % (prof[67] / (prof[71] / 100)))
                print("      group products, wave 0: ranges + load issue %.1f us, load wait %.1f us, products + stores %.1f us, whole routine %.1f us" % tuple(prof[68:72] / 100))
            for i, nm in enumerate(names):
                if prof[32 + i] > 0:
                    print(f"      {nm:12s} total {prof[i] / 100:9.1f} us  calls {int(prof[32 + i]):3d}  avg {prof[i] / 100 / prof[32 + i]:8.2f} us")
