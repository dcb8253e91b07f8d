"""ORACLE -- ctypes bindings of the CPU restatement (test infrastructure only).

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may
import this package; the product (rd_vio_amd) never does.

PARITY UNPINNED: the reference ships no fixtures and is unbuildable here; see
oracle/rdvio_oracle.h and DESIGN.md.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDVIO_ORACLE_LIB: load another build of the same sources (bench.py's cpu_baseline timing variants, see build_variant);
# the parity checker is always the default liboracle.so (-O2, no contraction, no fast-math)
_LIB_PATH = os.environ.get("RDVIO_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")

# CPU-baseline timing builds (SURVEY.md 8d / BASELINE.md section 3).  Timing only -- never used as the parity checker:
# -march=native ties the binary to the host it was compiled on, fast-math changes low bits.
VARIANTS = {
    # strong baseline: what the >= 200x target is quoted against
    "O3_native": ["-O3", "-march=native"],
    # the reference's own flags, /root/reference/CMakeLists.txt:10,17-18 (Debug: -Og; -msse -msse2 -msse3 -ffast-math -mtune=native)
    "Og_fastmath": ["-Og", "-msse", "-msse2", "-msse3", "-ffast-math", "-mtune=native"],
}


def _host_tag():
    """short hash of this host's CPU model + flags: a -march=native build must never run on another machine"""
    import hashlib

    try:
        txt = open("/proc/cpuinfo").read()
        key = "".join(l for l in txt.splitlines() if l.startswith(("model name", "flags")))[:20000]
    except OSError:
        key = "unknown"
    return hashlib.sha1(key.encode()).hexdigest()[:10]


def build_variant(name):
    """gcc the oracle sources with VARIANTS[name] into oracle/_build/ (on THIS host); returns the .so path"""
    flags = VARIANTS[name]
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"liboracle_{name}_{_host_tag()}.so")
    srcs = sorted(os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c"))
    deps = srcs + [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["gcc", "-std=c99", "-fPIC", "-shared", "-Wall", "-Wno-unused-parameter"] + flags + ["-o", out] + srcs + ["-lm"])
    return out



def build_backend(variant=None):
    """gcc oracle/backend/oracle_backend.c (the rdvio_backend function table over the oracle: the CPU path of the pipeline
    comparison) against liboracle.so -- or against the timing build `variant` (VARIANTS) -- into oracle/_build/; returns the
    loaded library (exports rdvio_oracle_backend_fill)."""
    root = os.path.dirname(_HERE)
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    if variant is None:
        build()
        core = os.path.join(_HERE, "liboracle.so")
        out = os.path.join(out_dir, "liboracle_backend.so")
        flags = ["-O2"]
    else:
        core = build_variant(variant)
        out = os.path.join(out_dir, f"liboracle_backend_{variant}_{_host_tag()}.so")
        flags = VARIANTS[variant]
    src = os.path.join(_HERE, "backend", "oracle_backend.c")
    deps = [src, core, os.path.join(root, "include", "rdvio_pipeline.h"), os.path.join(root, "include", "rdvio_hip.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["gcc", "-std=c99", "-fPIC", "-shared", "-Wall"] + flags + ["-o", out, src, core, "-Wl,-rpath," + os.path.dirname(core), "-lm"])
    return ctypes.CDLL(out)


STATE_SIZE = 16
PREINT_SIZE = 506
PREINT_T, PREINT_Q, PREINT_P, PREINT_V, PREINT_COV, PREINT_SIC, PREINT_JAC = 0, 1, 5, 8, 11, 236, 461


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    if os.environ.get("RDVIO_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in os.listdir(_HERE)
        if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a, t=ctypes.c_double):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------- lie algebra
def expmap(w):
    q = np.zeros(4)
    lib().ro_expmap(_p(_f64(w)), _p(q))
    return q


def logmap(q):
    w = np.zeros(3)
    lib().ro_logmap(_p(_f64(q)), _p(w))
    return w


def right_jacobian(w):
    J = np.zeros((3, 3))
    lib().ro_right_jacobian_c(_p(_f64(w)), _p(J))
    return J


def tangent_frame(z):
    T = np.zeros((3, 3))
    lib().ro_tangent_frame(_p(_f64(z)), _p(T))
    return T


def quat_plus(q, d):
    o = np.zeros(4)
    lib().ro_quat_plus(_p(_f64(q)), _p(_f64(d)), _p(o))
    return o


# ---------------------------------------------------------------- estimation
def preintegrate(imu, t_end, bg, ba, noise, jac=True, cov=True):
    imu = _f64(imu).reshape(-1, 7)
    out = np.zeros(PREINT_SIZE)
    lib().ro_preintegrate.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_double] + [ctypes.c_void_p] * 3 + [
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ok = lib().ro_preintegrate(len(imu), imu.ctypes.data, float(t_end), _f64(bg).ctypes.data, _f64(ba).ctypes.data,
                               _f64(noise).ctypes.data, int(jac), int(cov), out.ctypes.data)
    return out if ok else None


def preint_predict(pre, state_i):
    sj = np.zeros(16)
    lib().ro_preint_predict(_p(_f64(pre)), _p(_f64(state_i)), _p(sj))
    return sj


def reprojection_eval(tgt, ref, lm, tangent, z_ref, inv_depth, states, extr, W, jac=True):
    tgt, ref, lm = _i32(tgt), _i32(ref), _i32(lm)
    n = len(tgt)
    tangent, z_ref, inv_depth, states, extr, W = map(_f64, (tangent, z_ref, inv_depth, states, extr, W))
    r = np.zeros((n, 2))
    Jt = np.zeros((n, 2, 6)) if jac else None
    Jr = np.zeros((n, 2, 6)) if jac else None
    Jd = np.zeros((n, 2)) if jac else None
    lib().ro_reprojection_eval(ctypes.c_int(n), _p(tgt, ctypes.c_int32), _p(ref, ctypes.c_int32),
                               _p(lm, ctypes.c_int32), _p(tangent), _p(z_ref), _p(inv_depth), _p(states),
                               _p(extr), _p(W), _p(r), _p(Jt), _p(Jr), _p(Jd))
    return r, Jt, Jr, Jd


def rotation_prior_eval(q_tgt, q_ref, z_ref, tangent, extr, W):
    r = np.zeros(2)
    J = np.zeros((2, 3))
    lib().ro_rotation_prior_eval(_p(_f64(q_tgt)), _p(_f64(q_ref)), _p(_f64(z_ref)), _p(_f64(tangent)),
                                 _p(_f64(extr)), _p(_f64(W)), _p(r), _p(J))
    return r, J


def preintegration_eval(state_i, state_j, pre, bias_lin, extr, jac=True):
    r = np.zeros(15)
    Ji = np.zeros((15, 15)) if jac else None
    Jj = np.zeros((15, 15)) if jac else None
    lib().ro_preintegration_eval(_p(_f64(state_i)), _p(_f64(state_j)), _p(_f64(pre)), _p(_f64(bias_lin)),
                                 _p(_f64(extr)), _p(r), _p(Ji), _p(Jj))
    return r, Ji, Jj


def marginalization_eval(states, lin, S, f, jac=True):
    states = _f64(states).reshape(-1, 16)
    n = len(states)
    D = 15 * n
    r = np.zeros(D)
    J = np.zeros((D, D)) if jac else None
    lib().ro_marginalization_eval(ctypes.c_int(n), _p(states), _p(_f64(lin)), _p(_f64(S)), _p(_f64(f)), _p(r), _p(J))
    return r, J


class _MargProblem(ctypes.Structure):
    _fields_ = [
        ("nframes", ctypes.c_int), ("states", ctypes.c_void_p), ("extr", ctypes.c_void_p),
        ("sqrt_inv_cov", ctypes.c_void_p), ("np", ctypes.c_int), ("prior_frames", ctypes.c_void_p),
        ("lin", ctypes.c_void_p), ("S", ctypes.c_void_p), ("f", ctypes.c_void_p), ("preint01", ctypes.c_void_p),
        ("nfac", ctypes.c_int), ("tgt", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("lm", ctypes.c_void_p),
        ("tangent", ctypes.c_void_p), ("nlm", ctypes.c_int), ("z_ref", ctypes.c_void_p),
        ("inv_depth", ctypes.c_void_p),
    ]


def marginalize(states, extr, W, prior_frames, lin, S, f, preint01, tgt, ref, lm, tangent, z_ref, inv_depth):
    """CeresMarginalizationFactor::marginalize(0); returns S, f, lin, Lambda, eta."""
    states = _f64(states).reshape(-1, 16)
    nfm = len(states)
    keep = dict(states=states, extr=_f64(extr), W=_f64(W), pf=_i32(prior_frames), lin=_f64(lin), S=_f64(S),
                f=_f64(f), pre=_f64(preint01) if preint01 is not None else None, tgt=_i32(tgt), ref=_i32(ref),
                lm=_i32(lm), tangent=_f64(tangent), z_ref=_f64(z_ref), inv_depth=_f64(inv_depth))
    pb = _MargProblem()
    pb.nframes = nfm
    pb.states = keep["states"].ctypes.data
    pb.extr = keep["extr"].ctypes.data
    pb.sqrt_inv_cov = keep["W"].ctypes.data
    pb.np = len(keep["pf"])
    pb.prior_frames = keep["pf"].ctypes.data
    pb.lin = keep["lin"].ctypes.data
    pb.S = keep["S"].ctypes.data
    pb.f = keep["f"].ctypes.data
    pb.preint01 = keep["pre"].ctypes.data if keep["pre"] is not None else None
    pb.nfac = len(keep["tgt"])
    pb.tgt = keep["tgt"].ctypes.data
    pb.ref = keep["ref"].ctypes.data
    pb.lm = keep["lm"].ctypes.data
    pb.tangent = keep["tangent"].ctypes.data
    pb.nlm = len(keep["inv_depth"])
    pb.z_ref = keep["z_ref"].ctypes.data
    pb.inv_depth = keep["inv_depth"].ctypes.data
    R = 15 * (nfm - 1)
    S_out = np.zeros((R, R))
    f_out = np.zeros(R)
    lin_out = np.zeros((nfm - 1, 16))
    Lam = np.zeros((R, R))
    eta = np.zeros(R)
    lib().ro_marginalize(ctypes.byref(pb), _p(S_out), _p(f_out), _p(lin_out), _p(Lam), _p(eta))
    return S_out, f_out, lin_out, Lam, eta


# ---------------------------------------------------------------- image side
MAX_LEVELS = 4


class PyrLayout(ctypes.Structure):
    _fields_ = [
        ("levels", ctypes.c_int32), ("border", ctypes.c_int32),
        ("w", ctypes.c_int32 * MAX_LEVELS), ("h", ctypes.c_int32 * MAX_LEVELS),
        ("stride", ctypes.c_int32 * MAX_LEVELS),
        ("img_off", ctypes.c_int64 * MAX_LEVELS), ("deriv_off", ctypes.c_int64 * MAX_LEVELS),
        ("img_bytes", ctypes.c_int64), ("deriv_elems", ctypes.c_int64),
    ]


def pyr_layout(w, h, max_level=3):
    L = PyrLayout()
    lib().ro_pyr_layout_init(int(w), int(h), int(max_level), ctypes.byref(L))
    return L


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def clahe(img, clip=6.0, tiles_x=8, tiles_y=8):
    img = _u8(img)
    h, w = img.shape
    out = np.zeros_like(img)
    lib().ro_clahe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib().ro_clahe(img.ctypes.data, w, h, w, float(clip), tiles_x, tiles_y, out.ctypes.data, w)
    return out


def build_pyramid(img, max_level=3):
    """buildOpticalFlowPyramid of an (already CLAHE'd) image -> (layout, img arena u8, deriv arena i16)"""
    img = _u8(img)
    h, w = img.shape
    L = pyr_layout(w, h, max_level)
    pi = np.zeros(L.img_bytes, dtype=np.uint8)
    pd = np.zeros(L.deriv_elems, dtype=np.int16)
    lib().ro_build_pyramid(ctypes.c_void_p(img.ctypes.data), w, h, w, ctypes.byref(L),
                           ctypes.c_void_p(pi.ctypes.data), ctypes.c_void_p(pd.ctypes.data))
    return L, pi, pd


def preprocess(gray, clip=6.0, tiles_x=8, tiles_y=8, max_level=3):
    """OpenCvImage::preprocess: CLAHE then pyramid + Scharr."""
    gray = _u8(gray)
    h, w = gray.shape
    L = pyr_layout(w, h, max_level)
    pi = np.zeros(L.img_bytes, dtype=np.uint8)
    pd = np.zeros(L.deriv_elems, dtype=np.int16)
    lib().ro_preprocess.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib().ro_preprocess(gray.ctypes.data, w, h, w, float(clip), tiles_x, tiles_y, ctypes.addressof(L),
                        pi.ctypes.data, pd.ctypes.data)
    return L, pi, pd


def level_view(L, pi, lv, with_border=False):
    """numpy view of pyramid level lv of an image arena."""
    B, s = L.border, L.stride[lv]
    rows = L.h[lv] + 2 * B
    a = pi[L.img_off[lv]:L.img_off[lv] + s * rows].reshape(rows, s)
    return a[:, :L.w[lv] + 2 * B] if with_border else a[B:B + L.h[lv], B:B + L.w[lv]]


def deriv_view(L, pd, lv, with_border=False):
    B, s = L.border, L.stride[lv]
    rows = L.h[lv] + 2 * B
    a = pd[L.deriv_off[lv]:L.deriv_off[lv] + s * rows * 2].reshape(rows, s, 2)
    return a[:, :L.w[lv] + 2 * B] if with_border else a[B:B + L.h[lv], B:B + L.w[lv]]


def lk_flow(L, prev_img, prev_deriv, next_img, prev_xy, next_xy, max_iter=30, eps=0.01):
    prev_xy = np.ascontiguousarray(prev_xy, dtype=np.float32).reshape(-1, 2)
    nxt = np.ascontiguousarray(next_xy, dtype=np.float32).reshape(-1, 2).copy()
    n = len(prev_xy)
    st = np.zeros(n, dtype=np.uint8)
    lib().ro_lk_flow.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
    lib().ro_lk_flow(ctypes.addressof(L), prev_img.ctypes.data, prev_deriv.ctypes.data, next_img.ctypes.data, n,
                     prev_xy.ctypes.data, nxt.ctypes.data, st.ctypes.data, int(max_iter), float(eps))
    return nxt, st


def track_keypoints(L, cur, nxt, curr_pts, guess=None):
    """OpenCvImage::track_keypoints; cur/nxt are (img arena, deriv arena).  Returns next (n,2) double, status."""
    curr_pts = _f64(curr_pts).reshape(-1, 2)
    n = len(curr_pts)
    nx = _f64(guess).reshape(-1, 2).copy() if guess is not None else np.zeros((n, 2))
    st = np.zeros(n, dtype=np.uint8)
    lib().ro_track_keypoints.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                                  ctypes.c_int, ctypes.c_void_p]
    lib().ro_track_keypoints(ctypes.addressof(L), cur[0].ctypes.data, cur[1].ctypes.data, nxt[0].ctypes.data,
                             nxt[1].ctypes.data, n, curr_pts.ctypes.data, nx.ctypes.data,
                             int(guess is not None), st.ctypes.data)
    return nx, st


def harris_response(img, k=0.04):
    img = _u8(img)
    h, w = img.shape
    out = np.zeros((h, w), dtype=np.float32)
    lib().ro_harris_response.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                         ctypes.c_void_p]
    lib().ro_harris_response(img.ctypes.data, w, h, w, float(k), out.ctypes.data)
    return out


def good_features(img, max_corners, quality=1e-3, min_dist=20.0, k=0.04):
    img = _u8(img)
    h, w = img.shape
    cap = max_corners if max_corners > 0 else w * h
    xy = np.zeros((cap, 2), dtype=np.float32)
    resp = np.zeros(cap, dtype=np.float32)
    lib().ro_good_features.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                       ctypes.c_void_p]
    n = lib().ro_good_features(img.ctypes.data, w, h, w, int(max_corners), float(quality), float(min_dist), float(k),
                               xy.ctypes.data, resp.ctypes.data)
    return xy[:n].copy(), resp[:n].copy()


def detect_keypoints(img, existing, max_corners, min_dist):
    img = _u8(img)
    h, w = img.shape
    existing = _f64(existing).reshape(-1, 2)
    buf = np.zeros((len(existing) + max_corners, 2))
    buf[:len(existing)] = existing
    lib().ro_detect_keypoints.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_void_p, ctypes.c_int]
    n = lib().ro_detect_keypoints(img.ctypes.data, w, h, w, int(max_corners), float(min_dist), buf.ctypes.data,
                                  len(existing))
    return buf[:n].copy()


# ---------------------------------------------------------------- solver (A14)
class _BaProblem(ctypes.Structure):
    _fields_ = [
        ("n_frames", ctypes.c_int), ("frame_fixed", ctypes.c_void_p), ("extr", ctypes.c_void_p),
        ("sqrt_inv_cov", ctypes.c_void_p), ("n_landmarks", ctypes.c_int), ("lm_fixed", ctypes.c_void_p),
        ("z_ref", ctypes.c_void_p), ("n_factors", ctypes.c_int), ("tgt", ctypes.c_void_p), ("ref", ctypes.c_void_p),
        ("lm", ctypes.c_void_p), ("tangent", ctypes.c_void_p), ("n_rot", ctypes.c_int), ("rot_tgt", ctypes.c_void_p),
        ("rot_ref", ctypes.c_void_p), ("rot_zref", ctypes.c_void_p), ("rot_tangent", ctypes.c_void_p),
        ("n_preint", ctypes.c_int), ("pre_i", ctypes.c_void_p), ("pre_j", ctypes.c_void_p),
        ("preint", ctypes.c_void_p), ("np", ctypes.c_int), ("prior_frames", ctypes.c_void_p),
        ("lin", ctypes.c_void_p), ("S", ctypes.c_void_p), ("f", ctypes.c_void_p),
    ]


class BaSummary(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int), ("successful_steps", ctypes.c_int), ("initial_cost", ctypes.c_double),
                ("final_cost", ctypes.c_double), ("termination", ctypes.c_int)]


def pack_ba_problem(pb, struct_cls=_BaProblem):
    """dict -> (ctypes struct, keepalive).  Optional keys default to 'none'."""
    n = len(_f64(pb["states"]).reshape(-1, 16))
    nl = len(pb["inv_depth"])
    k = dict(
        frame_fixed=np.ascontiguousarray(pb.get("frame_fixed", np.zeros(n)), dtype=np.uint8),
        extr=_f64(pb["extr"]), W=_f64(pb["sqrt_inv_cov"]),
        lm_fixed=np.ascontiguousarray(pb.get("lm_fixed", np.zeros(nl)), dtype=np.uint8),
        z_ref=_f64(pb["z_ref"]).reshape(-1, 3), tgt=_i32(pb["tgt"]), ref=_i32(pb["ref"]), lm=_i32(pb["lm"]),
        tangent=_f64(pb["tangent"]).reshape(-1, 9),
        rot_tgt=_i32(pb.get("rot_tgt", [])), rot_ref=_i32(pb.get("rot_ref", [])),
        rot_zref=_f64(pb.get("rot_zref", np.zeros((0, 3)))), rot_tangent=_f64(pb.get("rot_tangent", np.zeros((0, 9)))),
        pre_i=_i32(pb.get("pre_i", [])), pre_j=_i32(pb.get("pre_j", [])),
        preint=_f64(pb.get("preint", np.zeros((0, PREINT_SIZE)))),
        prior_frames=_i32(pb.get("prior_frames", [])), lin=_f64(pb.get("lin", np.zeros((0, 16)))),
        S=_f64(pb.get("S", np.zeros((0, 0)))), f=_f64(pb.get("f", np.zeros(0))),
    )
    c = struct_cls()
    c.n_frames = n
    c.frame_fixed = k["frame_fixed"].ctypes.data
    c.extr = k["extr"].ctypes.data
    c.sqrt_inv_cov = k["W"].ctypes.data
    c.n_landmarks = nl
    c.lm_fixed = k["lm_fixed"].ctypes.data
    c.z_ref = k["z_ref"].ctypes.data
    c.n_factors = len(k["tgt"])
    c.tgt, c.ref, c.lm = k["tgt"].ctypes.data, k["ref"].ctypes.data, k["lm"].ctypes.data
    c.tangent = k["tangent"].ctypes.data
    c.n_rot = len(k["rot_tgt"])
    c.rot_tgt, c.rot_ref = k["rot_tgt"].ctypes.data, k["rot_ref"].ctypes.data
    c.rot_zref, c.rot_tangent = k["rot_zref"].ctypes.data, k["rot_tangent"].ctypes.data
    c.n_preint = len(k["pre_i"])
    c.pre_i, c.pre_j = k["pre_i"].ctypes.data, k["pre_j"].ctypes.data
    c.preint = k["preint"].ctypes.data
    c.np = len(k["prior_frames"])
    c.prior_frames = k["prior_frames"].ctypes.data
    c.lin, c.S, c.f = k["lin"].ctypes.data, k["S"].ctypes.data, k["f"].ctypes.data
    return c, k


def ba_solve(pb, max_iterations=30):
    """rdvio::Solver::solve on a problem dict -> (states, inv_depth, BaSummary)"""
    c, keep = pack_ba_problem(pb)
    states = _f64(pb["states"]).reshape(-1, 16).copy()
    invd = _f64(pb["inv_depth"]).copy()
    sm = BaSummary()
    lib().ro_ba_solve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib().ro_ba_solve(ctypes.addressof(c), int(max_iterations), states.ctypes.data, invd.ctypes.data,
                      ctypes.addressof(sm))
    return states, invd, sm
