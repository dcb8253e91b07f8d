"""Synthetic inputs of the shapes BASELINE.json names.

Used by tests/ and bench.py only -- there is no EuRoC data in the container or on
the GPU box, so every config runs on a seeded synthetic stream of the same
shape.  Pure numpy; nothing here touches the oracle or the HIP library.
"""
import numpy as np

GRAVITY = 9.80665

# configs/euroc_sensor.yaml:48-50 (camera -> body) and :10-12 (imu -> body, identity)
EUROC_EXTR = np.array([
    -7.7071797555374275e-03, 1.0499323370587278e-02, 7.0175280029197162e-01, 7.1230146066895372e-01,
    -0.0216401454975, -0.064676986768, 0.00981073058949,
    0.0, 0.0, 0.0, 1.0,
    0.0, 0.0, 0.0,
])
EUROC_K = np.array([[458.654, 0.0, 367.215], [0.0, 457.296, 248.375], [0.0, 0.0, 1.0]])
# configs/euroc_sensor.yaml:14-29
EUROC_NOISE = np.concatenate([
    (np.eye(3) * 2.8791302399999997e-08).ravel(), (np.eye(3) * 4.0e-6).ravel(),
    (np.eye(3) * 3.7608844899999997e-10).ravel(), (np.eye(3) * 9.0e-6).ravel()])


def sqrt_inv_cov_from_K(K, sigma2=0.5):
    """handler.cpp:117-119: frame->sqrt_inv_cov = K[0:2,0:2] / sqrt(cov)"""
    return np.array([[K[0, 0] / np.sqrt(sigma2), 0.0], [0.0, K[1, 1] / np.sqrt(sigma2)]])


# ------------------------------------------------------------------ small quaternion helpers (x,y,z,w)
def q_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def q_conj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def q_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def q_exp(w):
    n = np.linalg.norm(w)
    if n == 0:
        return np.array([0.0, 0.0, 0.0, 1.0])
    return np.concatenate([np.sin(0.5 * n) * w / n, [np.cos(0.5 * n)]])


def euler_to_q(roll, pitch, yaw):
    qx = q_exp(np.array([roll, 0, 0]))
    qy = q_exp(np.array([0, pitch, 0]))
    qz = q_exp(np.array([0, 0, yaw]))
    return q_mul(qz, q_mul(qy, qx))


def tangent_frame(z):
    """lie_algebra.cpp:47-56 + reprojection_factor.h:19-21, numpy twin used to build inputs."""
    d = int(np.argmax(np.abs(z)))
    e = np.zeros(3)
    e[(d + 1) % 3] = 1.0
    b1 = np.cross(z, e)
    b1 /= np.linalg.norm(b1)
    b2 = np.cross(z, b1)
    b2 /= np.linalg.norm(b2)
    return np.stack([b1, b2, z], axis=1)


# ------------------------------------------------------------------ trajectory
def traj_pose(t):
    p = np.array([1.5 * np.sin(0.4 * t), 1.0 * np.sin(0.6 * t), 0.3 * np.sin(0.9 * t)])
    q = euler_to_q(0.2 * np.sin(0.5 * t), 0.15 * np.sin(0.7 * t), 0.3 * np.sin(0.3 * t))
    return q, p


def traj_pose_rotation_phase(t, t_stop=2.2, tau=0.25):
    """the same trajectory whose translation smoothly comes to rest after t_stop while the rotation continues: a
    pure-rotation phase (FT_NO_TRANSLATION frames, rotation-prior factors, keyframe lifting in manage_keyframe)."""
    s = t if t <= t_stop else t_stop + tau * np.tanh((t - t_stop) / tau)
    _, p = traj_pose(s)
    q, _ = traj_pose(t)
    return q, p


def traj_vel(t, h=1e-5, pose_fn=None):
    pose_fn = pose_fn or traj_pose
    return (pose_fn(t + h)[1] - pose_fn(t - h)[1]) / (2 * h)


def traj_imu(t, h=1e-4, pose_fn=None):
    """analytic-derivative IMU sample (body frame): gyro, specific force."""
    pose_fn = pose_fn or traj_pose
    q, _ = pose_fn(t)
    qp, _ = pose_fn(t + h)
    qm, _ = pose_fn(t - h)
    dq = q_mul(q_conj(qm), qp)
    w = 2.0 * dq[:3] / (2 * h)
    acc_w = (pose_fn(t + h)[1] - 2 * pose_fn(t)[1] + pose_fn(t - h)[1]) / (h * h)
    a = q_to_mat(q).T @ (acc_w - np.array([0, 0, -GRAVITY]))
    return w, a


TRUE_BG = np.array([1e-3, -2e-3, 5e-4])
TRUE_BA = np.array([0.02, -0.01, 0.03])


def make_imu_segment(t0, t1, rate=200.0, rng=None, noise=True, bg=None, ba=None, pose_fn=None):
    """IMU samples in [t0, t1) as n x 7 (t, w, a); optional constant sensor biases are added."""
    ts = np.arange(t0, t1 - 1e-9, 1.0 / rate)
    out = np.zeros((len(ts), 7))
    for i, t in enumerate(ts):
        w, a = traj_imu(t, pose_fn=pose_fn)
        if noise and rng is not None:
            w = w + rng.normal(0, np.sqrt(2.8791302399999997e-08 * rate), 3)
            a = a + rng.normal(0, np.sqrt(4.0e-6 * rate), 3)
        if bg is not None:
            w = w + bg
        if ba is not None:
            a = a + ba
        out[i] = np.concatenate([[t], w, a])
    return out


# ------------------------------------------------------------------ BA problem
def make_ba_problem(n_frames=9, n_landmarks=150, seed=648, K=EUROC_K, extr=EUROC_EXTR, pix_noise=0.5,
                    state_noise=True, dt_frame=0.25, obs_prob=0.9, t0=1.0):
    """A window of `n_frames` keyframes observing `n_landmarks` landmarks.

    Returns a dict of plain arrays in the layouts of include/rdvio_hip.h:
    states (n,16), extr (14), sqrt_inv_cov (2,2), z_ref (L,3), inv_depth (L), and the
    factor list tgt/ref/lm (F) + tangent (F,9).  Landmarks are anchored in the first
    frame that observes them (track.h first_keypoint); every other observation is a factor
    (sliding_window_tracker.cpp:261-275).
    """
    rng = np.random.default_rng(seed)
    qcs, pcs = extr[0:4], extr[4:7]
    Rcs = q_to_mat(qcs)
    states = np.zeros((n_frames, 16))
    cams = []
    for i in range(n_frames):
        t = t0 + dt_frame * i
        q, p = traj_pose(t)
        states[i, 0:4] = q
        states[i, 4:7] = p
        states[i, 7:10] = traj_vel(t)
        states[i, 10:13] = TRUE_BG
        states[i, 13:16] = TRUE_BA
        Rwc = q_to_mat(q) @ Rcs
        pwc = p + q_to_mat(q) @ pcs
        cams.append((Rwc, pwc))
    w_img, h_img = 2 * K[0, 2], 2 * K[1, 2]
    z_ref, inv_depth, tgt, ref, lm, tangent = [], [], [], [], [], []
    l = 0
    attempts = 0
    while l < n_landmarks and attempts < 100 * n_landmarks:
        attempts += 1
        # sample a landmark in front of a random camera
        c = rng.integers(0, n_frames)
        Rwc, pwc = cams[c]
        u = np.array([rng.uniform(30, w_img - 30), rng.uniform(30, h_img - 30)])
        depth = 1.0 / rng.uniform(0.1, 1.0)
        ray = np.array([(u[0] - K[0, 2]) / K[0, 0], (u[1] - K[1, 2]) / K[1, 1], 1.0])
        X = pwc + Rwc @ (ray / np.linalg.norm(ray) * depth)
        obs = []
        for i, (R, p) in enumerate(cams):
            y = R.T @ (X - p)
            if y[2] < 0.2:
                continue
            px = np.array([K[0, 0] * y[0] / y[2] + K[0, 2], K[1, 1] * y[1] / y[2] + K[1, 2]])
            if not (20 <= px[0] < w_img - 20 and 20 <= px[1] < h_img - 20):
                continue
            if rng.uniform() > obs_prob:
                continue
            px = px + rng.normal(0, pix_noise, 2)
            b = np.array([(px[0] - K[0, 2]) / K[0, 0], (px[1] - K[1, 2]) / K[1, 1], 1.0])
            obs.append((i, b / np.linalg.norm(b)))
        if len(obs) < 2:
            continue
        a_frame, a_bearing = obs[0]
        Ra, pa = cams[a_frame]
        z_ref.append(a_bearing)
        inv_depth.append(1.0 / np.linalg.norm(Ra.T @ (X - pa)))
        for (i, b) in obs[1:]:
            tgt.append(i)
            ref.append(a_frame)
            lm.append(l)
            tangent.append(tangent_frame(b).ravel())
        l += 1
    out = dict(
        states=states, extr=extr.copy(), sqrt_inv_cov=sqrt_inv_cov_from_K(K), K=K.copy(),
        z_ref=np.array(z_ref), inv_depth=np.array(inv_depth),
        tgt=np.array(tgt, dtype=np.int32), ref=np.array(ref, dtype=np.int32), lm=np.array(lm, dtype=np.int32),
        tangent=np.array(tangent), states_true=states.copy(), inv_depth_true=np.array(inv_depth))
    if state_noise:
        # perturb what BA is supposed to recover
        for i in range(1, n_frames):
            out["states"][i, 0:4] = q_mul(states[i, 0:4], q_exp(rng.normal(0, 2e-3, 3)))
            out["states"][i, 0:4] /= np.linalg.norm(out["states"][i, 0:4])
            out["states"][i, 4:7] += rng.normal(0, 1e-2, 3)
            out["states"][i, 7:10] += rng.normal(0, 1e-2, 3)
        out["states"][:, 10:13] += rng.normal(0, 1e-5, (n_frames, 3))
        out["states"][:, 13:16] += rng.normal(0, 1e-4, (n_frames, 3))
        out["inv_depth"] = out["inv_depth"] * rng.uniform(0.9, 1.1, len(inv_depth))
    return out


# ------------------------------------------------------------------ synthetic imagery
def _value_noise(xs, ys, rng_seed, cell):
    """smooth value noise: random lattice values with smoothstep interpolation, evaluated at (xs, ys)."""
    gx = xs / cell
    gy = ys / cell
    x0 = np.floor(gx).astype(np.int64)
    y0 = np.floor(gy).astype(np.int64)
    fx = gx - x0
    fy = gy - y0
    fx = fx * fx * (3 - 2 * fx)
    fy = fy * fy * (3 - 2 * fy)

    def lat(ix, iy):
        # hash lattice coordinates -> [0,1)
        hsh = (ix * 73856093) ^ (iy * 19349663) ^ (rng_seed * 83492791)
        hsh = (hsh ^ (hsh >> 13)) * 1274126177
        hsh = hsh ^ (hsh >> 16)
        return (hsh & 0xFFFFFF) / float(0x1000000)

    v00, v10 = lat(x0, y0), lat(x0 + 1, y0)
    v01, v11 = lat(x0, y0 + 1), lat(x0 + 1, y0 + 1)
    return (v00 * (1 - fx) + v10 * fx) * (1 - fy) + (v01 * (1 - fx) + v11 * fx) * fy


def render_scene(w, h, offset=(0.0, 0.0), seed=648, n_blobs=None, rot=0.0):
    """u8 image of a fixed planar 'world' texture sampled at pixel + offset (optionally rotated about
    the image centre by `rot` rad): 3-octave value noise (mean 110, amplitude 40) + Gaussian blobs
    (sigma in U[1.5,3], amplitude in U[60,200])."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    c, s = np.cos(rot), np.sin(rot)
    X = c * (xs - cx) - s * (ys - cy) + cx + offset[0]
    Y = s * (xs - cx) + c * (ys - cy) + cy + offset[1]
    img = np.full((h, w), 110.0)
    for o, cell in enumerate((64.0, 32.0, 16.0)):
        img += (40.0 / (1.75 * (1 << o)) * 2.0) * (_value_noise(X, Y, seed + o, cell) - 0.5) * 2.0
    if n_blobs is None:
        n_blobs = max(50, w * h // 600)
    bx = rng.uniform(-40, w + 40, n_blobs)
    by = rng.uniform(-40, h + 40, n_blobs)
    bs = rng.uniform(1.5, 3.0, n_blobs)
    ba = rng.uniform(60, 200, n_blobs) * rng.choice([-0.5, 1.0], n_blobs)
    for i in range(n_blobs):
        r = int(4 * bs[i] + 2)
        # bounding box of the blob in image coordinates (approximate for small rot/offset)
        ux = bx[i] - offset[0]
        uy = by[i] - offset[1]
        x0, x1 = int(max(0, ux - r - 8)), int(min(w, ux + r + 9))
        y0, y1 = int(max(0, uy - r - 8)), int(min(h, uy + r + 9))
        if x0 >= x1 or y0 >= y1:
            continue
        d2 = (X[y0:y1, x0:x1] - bx[i]) ** 2 + (Y[y0:y1, x0:x1] - by[i]) ** 2
        img[y0:y1, x0:x1] += ba[i] * np.exp(-d2 / (2 * bs[i] ** 2))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def jittered_grid(w, h, nx, ny, seed=648, margin=40):
    """nx*ny feature points on a jittered grid inside the image (SURVEY.md 8d kernel-level LK input)."""
    rng = np.random.default_rng(seed)
    gx = np.linspace(margin, w - margin, nx)
    gy = np.linspace(margin, h - margin, ny)
    pts = np.stack(np.meshgrid(gx, gy), axis=-1).reshape(-1, 2)
    pts += rng.uniform(-3, 3, pts.shape)
    return pts


def make_window_problem(n_frames=9, n_landmarks=150, seed=648, with_prior=True, with_preint=True,
                        preintegrate=None, **kw):
    """make_ba_problem + what refine_window adds (sliding_window_tracker.cpp:226-300): a preintegration
    factor between consecutive frames and a marginalisation prior over all frames but the newest
    (marginalization_factor.h:14-31: frame 0 pose pinned with 1e15).  `preintegrate(imu, t_end, bg, ba)`
    supplies the PreIntegrator (the oracle's or the HIP one) so this module stays independent of both."""
    pb = make_ba_problem(n_frames=n_frames, n_landmarks=n_landmarks, seed=seed, **kw)
    rng = np.random.default_rng(seed + 1)
    dt_frame, t0 = kw.get("dt_frame", 0.25), kw.get("t0", 1.0)
    if with_preint:
        pre, pi, pj = [], [], []
        for j in range(1, n_frames):
            imu = make_imu_segment(t0 + dt_frame * (j - 1), t0 + dt_frame * j, rng=rng, bg=TRUE_BG, ba=TRUE_BA)
            b = pb["states"][j - 1, 10:16]
            pre.append(preintegrate(imu, t0 + dt_frame * j, b[:3], b[3:]))
            pi.append(j - 1)
            pj.append(j)
        pb["preint"] = np.array(pre)
        pb["pre_i"] = np.array(pi, dtype=np.int32)
        pb["pre_j"] = np.array(pj, dtype=np.int32)
    if with_prior:
        npf = n_frames - 1
        D = 15 * npf
        S = np.zeros((D, D))
        S[0:6, 0:6] = 1e15 * np.eye(6)
        pb["prior_frames"] = np.arange(npf, dtype=np.int32)
        pb["lin"] = pb["states"][:npf].copy()
        pb["S"] = S
        pb["f"] = np.zeros(D)
    pb["frame_fixed"] = np.zeros(n_frames, dtype=np.uint8)
    pb["lm_fixed"] = np.zeros(len(pb["inv_depth"]), dtype=np.uint8)
    return pb


def add_rotation_priors(pb, n_rot=40, tgt=None, seed=77, K=EUROC_K, pix_noise=0.5):
    """Rotation-prior factors (CeresRotationPriorFactor, ceres/rotation_factor.h:11-67) as refine_subwindow adds them for
    the valid but untriangulated tracks of the last subframe (sliding_window_tracker.cpp:389-404): per factor the target
    frame, the frame of the track's first observation, that observation's bearing (rot_zref) and the tangent frame
    [b1 b2 z] of the bearing observed in the target frame.  The 3-D points are far away (bearing-only information)."""
    rng = np.random.default_rng(seed)
    n = len(pb["states"])
    tgt = n - 1 if tgt is None else tgt
    ex = pb["extr"]
    Rcs, pcs = q_to_mat(ex[0:4]), ex[4:7]
    st = pb.get("states_true", pb["states"])

    def cam(i):
        R = q_to_mat(st[i, 0:4])
        return R @ Rcs, st[i, 4:7] + R @ pcs
    w_img, h_img = 2 * K[0, 2], 2 * K[1, 2]
    rt, rr, rz, rT = [], [], [], []
    tries = 0
    while len(rt) < n_rot and tries < 100 * n_rot:
        tries += 1
        ref = int(rng.integers(0, n))
        if ref == tgt:
            continue
        Rr, pr = cam(ref)
        u = np.array([rng.uniform(40, w_img - 40), rng.uniform(40, h_img - 40)])
        ray = np.array([(u[0] - K[0, 2]) / K[0, 0], (u[1] - K[1, 2]) / K[1, 1], 1.0])
        X = pr + Rr @ (ray / np.linalg.norm(ray) * rng.uniform(15.0, 60.0))
        Rt, pt = cam(tgt)
        y = Rt.T @ (X - pt)
        if y[2] < 0.5:
            continue
        px = np.array([K[0, 0] * y[0] / y[2] + K[0, 2], K[1, 1] * y[1] / y[2] + K[1, 2]])
        if not (20 <= px[0] < w_img - 20 and 20 <= px[1] < h_img - 20):
            continue
        px = px + rng.normal(0, pix_noise, 2)
        b = np.array([(px[0] - K[0, 2]) / K[0, 0], (px[1] - K[1, 2]) / K[1, 1], 1.0])
        rt.append(tgt)
        rr.append(ref)
        rz.append(ray / np.linalg.norm(ray))
        rT.append(tangent_frame(b / np.linalg.norm(b)).ravel())
    pb["rot_tgt"] = np.array(rt, dtype=np.int32)
    pb["rot_ref"] = np.array(rr, dtype=np.int32)
    pb["rot_zref"] = np.array(rz).reshape(-1, 3)
    pb["rot_tangent"] = np.array(rT).reshape(-1, 9)
    return pb


def make_marg_inputs(pb, with_full_prior=False, seed=9):
    """What Map::marginalize_frame(0) hands to marginalize() for a window problem: the current prior, the
    preintegration between frames 0 and 1 and the reprojection factors of the tracks the victim (frame 0)
    observes (marginalization_factor.h:233-380).  Returns the positional argument tuple shared by the oracle's and
    the HIP binding's marginalize(): (states, extr, W, prior_frames, lin, S, f, preint01, tgt, ref, lm, tangent,
    z_ref, inv_depth)."""
    rng = np.random.default_rng(seed)
    n = len(pb["states"])
    npf = n - 1
    D = 15 * npf
    if with_full_prior:
        S = np.linalg.qr(rng.normal(size=(D, D)) * 3.0)[1]
        f = rng.normal(size=D)
        lin = pb["states"][:npf].copy()
        lin[:, 4:7] += rng.normal(0, 1e-3, (npf, 3))
    else:
        S, f, lin = pb["S"], pb["f"], pb["lin"]
    seen = set(pb["lm"][(pb["tgt"] == 0) | (pb["ref"] == 0)].tolist())
    keep = np.array([l in seen for l in pb["lm"]], dtype=bool)
    return (pb["states"], pb["extr"], pb["sqrt_inv_cov"], np.arange(npf, dtype=np.int32), lin, S, f, pb["preint"][0],
            pb["tgt"][keep], pb["ref"][keep], pb["lm"][keep], pb["tangent"][keep], pb["z_ref"], pb["inv_depth"])


def steady_state_marg_inputs(pb, marginalize):
    """make_marg_inputs with the prior a running system hands to marginalize(): the first marginalisation of a session
    starts from the initial prior (frame-0 pose pin, rank-deficient on the other frames); every later one starts from
    the prior the previous marginalisation left behind -- pose information on every retained frame, velocity / bias
    information on the oldest.  That structure is produced here by marginalising once (`marginalize` = the HIP
    binding's or the oracle's marginalize(*args) -> (S, f, lin, ...)) and re-attaching the result to the window's
    first frames."""
    first = make_marg_inputs(pb)
    S1, f1 = marginalize(*first)[:2]
    args = list(first)
    npf = len(pb["states"]) - 1
    args[4] = pb["states"][:npf].copy()   # lin: the prior is linearised at the current states
    args[5] = np.ascontiguousarray(S1)
    args[6] = np.ascontiguousarray(f1)
    return tuple(args)


# ------------------------------------------------------------------ 3-D consistent synthetic stream (pipeline tests)
ROOM_HALF = np.array([6.0, 6.0, 2.5])   # the 12 x 12 x 5 m box of SURVEY.md 8d (config 5)


def _hash01(ix, iy, seed):
    hsh = (ix * 73856093) ^ (iy * 19349663) ^ (seed * 83492791)
    hsh = (hsh ^ (hsh >> 13)) * 1274126177
    hsh = hsh ^ (hsh >> 16)
    return (hsh & 0xFFFFFF) / float(0x1000000)


def _wall_texture(u, v, seed):
    """grey value of a wall at plane coordinates (u, v) in metres: 4-octave value noise + a jittered lattice of
    Gaussian blobs (one per 0.3 m cell, sigma 2-4 cm) -- corner-rich at 2-6 m viewing distance."""
    img = np.full(u.shape, 110.0)
    for o, cell in enumerate((1.2, 0.6, 0.3, 0.15)):
        img += (80.0 / (1.75 * (1 << o))) * (_value_noise(u, v, seed + o, cell) - 0.5) * 2.0
    cell = 0.3
    cu = np.floor(u / cell).astype(np.int64)
    cv = np.floor(v / cell).astype(np.int64)
    for du in (-1, 0, 1):
        for dv in (-1, 0, 1):
            ix, iy = cu + du, cv + dv
            bx = (ix + _hash01(ix, iy, seed + 11)) * cell
            by = (iy + _hash01(ix, iy, seed + 12)) * cell
            sg = 0.02 + 0.02 * _hash01(ix, iy, seed + 13)
            am = (60.0 + 140.0 * _hash01(ix, iy, seed + 14)) * np.where(_hash01(ix, iy, seed + 15) < 0.35, -0.5, 1.0)
            img += am * np.exp(-((u - bx) ** 2 + (v - by) ** 2) / (2 * sg * sg))
    return img


MOVER_START = 2.6   # [s] the billboard rests until then (its points get mapped), then moves on its own


def mover_center(t):
    """centre of the billboard of render_room(..., mover_t=t): a 0.8 m textured square at height 1.6 m that is part of the
    static scene until MOVER_START and then drifts sideways at 0.3 m/s -- an independently moving rigid object whose
    landmarks are already in the map (the dynamic-outlier scenario of the RD path)."""
    dt = max(0.0, t - MOVER_START)
    return np.array([1.1 - 0.25 * dt, 0.8 - 0.2 * dt, 1.6])


def render_room(q_wb, p_wb, K, w, h, extr=EUROC_EXTR, seed=648, mover_t=None):
    """u8 image seen by the camera of a body at pose (q_wb, p_wb) inside the textured box room: per-pixel ray cast
    against the six walls (geometrically consistent across frames: true parallax and perspective).  mover_t adds a
    horizontal textured square at mover_center(mover_t) in front of the ceiling."""
    q_wc = q_mul(q_wb, extr[0:4])
    R = q_to_mat(q_wc)
    c = p_wb + q_to_mat(q_wb) @ extr[4:7]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d_c = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], axis=-1)
    d = d_c @ R.T
    best_t = np.full((h, w), np.inf)
    img = np.zeros((h, w))
    plane = 0
    for axis in range(3):
        for sign in (-1.0, 1.0):
            plane += 1
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (sign * ROOM_HALF[axis] - c[axis]) / d[..., axis]
            ok = (t > 1e-6) & (t < best_t)
            hit = c[None, None, :] + t[..., None] * d
            a1, a2 = [a for a in range(3) if a != axis]
            ok &= (np.abs(hit[..., a1]) <= ROOM_HALF[a1] + 1e-9) & (np.abs(hit[..., a2]) <= ROOM_HALF[a2] + 1e-9)
            if not ok.any():
                continue
            tex = _wall_texture(hit[..., a1][ok], hit[..., a2][ok], seed + 100 * plane)
            img[ok] = tex
            best_t[ok] = t[ok]
    if mover_t is not None:
        mc = mover_center(mover_t)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (mc[2] - c[2]) / d[..., 2]
        hit = c[None, None, :] + t[..., None] * d
        u, v = hit[..., 0] - mc[0], hit[..., 1] - mc[1]
        ok = (t > 1e-6) & (t < best_t) & (np.abs(u) <= 0.4) & (np.abs(v) <= 0.4)
        if ok.any():
            # feathered border: a hard step edge would own GFTT's quality threshold (1e-3 of the strongest corner)
            alpha = np.clip((0.4 - np.maximum(np.abs(u[ok]), np.abs(v[ok]))) / 0.08, 0.0, 1.0)
            img[ok] = alpha * _wall_texture(u[ok], v[ok], seed + 4242) + (1.0 - alpha) * img[ok]   # the texture travels with the object
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _render_job(job):
    (q, p), K, w, h, seed, mover_t = job
    return render_room(q, p, K, w, h, seed=seed, mover_t=mover_t)


def _render_frames(jobs, workers=None):
    """render_room for every job; frames are independent pure functions of their pose, so long streams are rendered by a
    pool of forked workers (numpy only in the children; results do not depend on the number of workers)."""
    import os

    if workers is None:
        workers = int(os.environ.get("RDVIO_SYNTH_WORKERS", min(8, os.cpu_count() or 1)))
    if workers <= 1 or len(jobs) < 8:
        return np.stack([_render_job(j) for j in jobs])
    import multiprocessing as mp

    with mp.get_context("fork").Pool(workers) as pool:
        return np.stack(pool.map(_render_job, jobs, chunksize=1))


def make_stream(n_frames, w=752, h=480, K=EUROC_K, t0=1.0, cam_rate=20.0, imu_rate=200.0, seed=648, imu_noise=True, pose_fn=None,
                mover=False, workers=None):
    """A synthetic EuRoC-shaped stream on the SURVEY.md 8d trajectory: images (n_frames x h x w u8), frame times, IMU rows
    (t, gyro, acc) covering the frames with the constant biases TRUE_BG / TRUE_BA added, and the ground-truth body
    states at the frame times as rows (t, q, p, v, bg, ba)."""
    rng = np.random.default_rng(seed + 1)
    ts = t0 + np.arange(n_frames) / cam_rate
    pose_fn = pose_fn or traj_pose
    frames = _render_frames([(pose_fn(t), K, w, h, seed, (t if mover else None)) for t in ts], workers)
    imu = make_imu_segment(t0 - 0.5 / imu_rate - 2.0 / imu_rate, ts[-1] + 3.0 / imu_rate, rate=imu_rate, rng=rng if imu_noise else None,
                           noise=imu_noise, bg=TRUE_BG, ba=TRUE_BA, pose_fn=pose_fn)
    gt = np.zeros((n_frames, 17))
    for i, t in enumerate(ts):
        q, p = pose_fn(t)
        gt[i] = np.concatenate([[t], q, p, traj_vel(t, pose_fn=pose_fn), TRUE_BG, TRUE_BA])
    return frames, ts, imu, gt
