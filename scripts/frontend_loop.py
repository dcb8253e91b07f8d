"""Developer timing aid: the feature tracker's device calls in a loop on one EuRoC-sized image pair (GPU box).  Run under
`rocprofv3 --kernel-trace --stats` for per-kernel durations; prints the wall time per call itself."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import rd_vio_amd
from rd_vio_amd import synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (752, 480)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 150
REPS = 200
ctx = rd_vio_amd.Context(max_width=W, max_height=H, max_features=4096)
a = synth.render_scene(W, H, seed=11)
b = synth.render_scene(W, H, seed=11, offset=(3.3, -2.1), rot=0.004)
ga, gb = rd_vio_amd.HipImage(ctx, 0, a), rd_vio_amd.HipImage(ctx, 1, b)
ga.preprocess(); gb.preprocess()
pts = ga.detect_keypoints(np.zeros((0, 2)), max_points=N, keypoint_distance=20.0)
guess = pts - np.array([3.3, -2.1])


def timeit(name, fn):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    ctx.sync()
    print(f"{name:28s} {1e6 * (time.perf_counter() - t0) / REPS:8.1f} us per call")


print(f"{W}x{H}, {len(pts)} keypoints")
timeit("preprocess (upload + kernels)", lambda: ga.preprocess())
timeit("track_keypoints", lambda: ga.track_keypoints(gb, pts, guess))
timeit("detect_keypoints", lambda: ga.detect_keypoints(pts[: len(pts) // 2], max_points=N, keypoint_distance=20.0))
nxt, st = ga.track_keypoints(gb, pts, guess)
print("tracked", int(st.sum()), "of", len(pts))
import ctypes
st = (ctypes.c_int32 * 5)()
ctx._lib.rdvio_hip_debug_last_select_stamps(ctx._h, st)
print("gftt_select_kernel stamps (us since kernel start): load %.1f sort %.1f cell lists %.1f greedy %.1f; %d candidates, path %d" % (st[0] / 100, st[1] / 100, st[2] / 100, st[3] / 100, st[4], ctx._lib.rdvio_hip_debug_last_select_path(ctx._h)))
