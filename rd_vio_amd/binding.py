"""ctypes binding of librdvio_hip.so (include/rdvio_hip.h).  No compute happens in Python."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATE_SIZE = 16
PREINT_SIZE = 506
MAX_LEVELS = 4

# every symbol include/rdvio_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "rdvio_hip_version", "rdvio_hip_pyr_layout_init", "rdvio_hip_ctx_create", "rdvio_hip_ctx_destroy",
    "rdvio_hip_last_error", "rdvio_hip_sync", "rdvio_hip_ctx_set_lane_stream", "rdvio_hip_ctx_set_wait_mode", "rdvio_hip_lane_wait", "rdvio_hip_lane_sync", "rdvio_hip_image_preprocess", "rdvio_hip_image_upload", "rdvio_hip_image_preprocess_uploaded", "rdvio_hip_image_preprocess_dev",
    "rdvio_hip_image_download", "rdvio_hip_track_keypoints", "rdvio_hip_track_keypoints_dev", "rdvio_hip_lk_flow",
    "rdvio_hip_detect_keypoints", "rdvio_hip_harris_response", "rdvio_hip_image_release", "rdvio_hip_preintegrate",
    "rdvio_hip_preintegrate_dev", "rdvio_hip_preintegrate_estimator", "rdvio_hip_preintegrate_estimator_begin", "rdvio_hip_preintegrate_estimator_end", "rdvio_hip_ctx_attach_thread", "rdvio_hip_ctx_ensure_lane_streams",
    "rdvio_hip_reprojection_eval", "rdvio_hip_rotation_prior_eval", "rdvio_hip_ba_solve", "rdvio_hip_ba_upload", "rdvio_hip_ba_solve_resident",
    "rdvio_hip_ba_fetch", "rdvio_hip_ba_fetch_enqueue", "rdvio_hip_ba_upload_chained", "rdvio_hip_ba_linearize", "rdvio_hip_ctx_team_retries", "rdvio_hip_debug_last_select_path", "rdvio_hip_debug_last_select_stamps", "rdvio_hip_ctx_set_kernel_timing", "rdvio_hip_ctx_get_kernel_timing", "rdvio_hip_marginalize", "rdvio_hip_marginalize_upload", "rdvio_hip_marginalize_resident",
    "rdvio_hip_marginalize_fetch", "rdvio_hip_parsac_score", "rdvio_hip_parsac_generate_score", "rdvio_hip_parsac_fetch", "rdvio_hip_ransac_generate_score", "rdvio_hip_ransac_fetch", "rdvio_hip_thin_tracks",
    "rdvio_hip_frame_step", "rdvio_hip_run_sequences",
]


class RdvioError(RuntimeError):
    """code: the RDVIO_ERR_* value; summary: the solver summary when the failing call filled one (RDVIO_ERR_TIMEOUT)"""
    code = None
    summary = None


ERR_TIMEOUT = 4
LANE_FRONTEND, LANE_SOLVER, LANE_MARG = 0, 1, 2


class PyrLayout(ctypes.Structure):
    _fields_ = [
        ("levels", ctypes.c_int32), ("border", ctypes.c_int32),
        ("w", ctypes.c_int32 * MAX_LEVELS), ("h", ctypes.c_int32 * MAX_LEVELS),
        ("stride", ctypes.c_int32 * MAX_LEVELS),
        ("img_off", ctypes.c_int64 * MAX_LEVELS), ("deriv_off", ctypes.c_int64 * MAX_LEVELS),
        ("img_bytes", ctypes.c_int64), ("deriv_elems", ctypes.c_int64),
    ]


class FrameStep(ctypes.Structure):
    """rdvio_frame_step (include/rdvio_hip.h): one camera frame of the resident hot path / one sequence of the multi-sequence driver"""
    _fields_ = [
        ("ctx", ctypes.c_void_p),
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("stride", ctypes.c_int32), ("n_images", ctypes.c_int32),
        ("images_dev", ctypes.POINTER(ctypes.c_void_p)),
        ("n_features", ctypes.c_int32), ("keypoints_capacity", ctypes.c_int32),
        ("curr_xy_dev", ctypes.c_void_p), ("next_xy_dev", ctypes.c_void_p), ("status_dev", ctypes.c_void_p),
        ("keypoints_host", ctypes.c_void_p), ("min_distance", ctypes.c_double),
        ("nseg", ctypes.c_int32), ("ba_iterations", ctypes.c_int32),
        ("seg_off_dev", ctypes.c_void_p), ("imu_dev", ctypes.c_void_p), ("par_dev", ctypes.c_void_p), ("noise_dev", ctypes.c_void_p),
        ("preint_out_dev", ctypes.c_void_p),
        ("overlap", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class BaProblem(ctypes.Structure):
    """rdvio_ba_problem (include/rdvio_hip.h)"""
    _fields_ = [
        ("n_frames", ctypes.c_int32), ("states", ctypes.c_void_p), ("frame_fixed", ctypes.c_void_p),
        ("extr", ctypes.c_void_p), ("sqrt_inv_cov", ctypes.c_void_p),
        ("n_landmarks", ctypes.c_int32), ("z_ref", ctypes.c_void_p), ("inv_depth", ctypes.c_void_p),
        ("lm_fixed", ctypes.c_void_p),
        ("n_factors", ctypes.c_int32), ("tgt", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("lm", ctypes.c_void_p),
        ("tangent", ctypes.c_void_p),
        ("n_rot", ctypes.c_int32), ("rot_tgt", ctypes.c_void_p), ("rot_ref", ctypes.c_void_p),
        ("rot_zref", ctypes.c_void_p), ("rot_tangent", ctypes.c_void_p),
        ("n_preint", ctypes.c_int32), ("pre_i", ctypes.c_void_p), ("pre_j", ctypes.c_void_p),
        ("preint", ctypes.c_void_p),
        ("n_prior", ctypes.c_int32), ("prior_frames", ctypes.c_void_p), ("prior_lin", ctypes.c_void_p),
        ("prior_S", ctypes.c_void_p), ("prior_f", ctypes.c_void_p),
        ("n_pre_jobs", ctypes.c_int32), ("job_seg_off", ctypes.c_void_p), ("job_imu", ctypes.c_void_p), ("job_par", ctypes.c_void_p),
        ("job_noise", ctypes.c_void_p), ("job_preint_out", ctypes.c_void_p),
    ]


class MargProblem(ctypes.Structure):
    """rdvio_marg_problem (include/rdvio_hip.h)"""
    _fields_ = [
        ("n_frames", ctypes.c_int32), ("states", ctypes.c_void_p), ("extr", ctypes.c_void_p),
        ("sqrt_inv_cov", ctypes.c_void_p), ("n_prior", ctypes.c_int32), ("prior_frames", ctypes.c_void_p),
        ("prior_lin", ctypes.c_void_p), ("prior_S", ctypes.c_void_p), ("prior_f", ctypes.c_void_p),
        ("preint01", ctypes.c_void_p), ("n_landmarks", ctypes.c_int32), ("z_ref", ctypes.c_void_p),
        ("inv_depth", ctypes.c_void_p), ("n_factors", ctypes.c_int32), ("tgt", ctypes.c_void_p),
        ("ref", ctypes.c_void_p), ("lm", ctypes.c_void_p), ("tangent", ctypes.c_void_p),
    ]


class BaLinearization(ctypes.Structure):
    """rdvio_ba_linearization (include/rdvio_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("r_preint", "J_preint", "r_prior", "J_prior", "H", "g", "lm_info", "lm_grad", "S_reduced",
                                               "c_reduced")] + [("N", ctypes.c_int32)]


class BaSummary(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("successful_steps", ctypes.c_int32),
                ("initial_cost", ctypes.c_double), ("final_cost", ctypes.c_double), ("termination", ctypes.c_int32)]


def lib_path():
    return os.path.join(_HERE, "librdvio_hip.so")


def load_library():
    """Load librdvio_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RdvioError(f"{path} not built: run `python -m rd_vio_amd.build` (or __graft_entry__.build())")
    try:
        # share torch's HIP runtime when torch is present in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the binding itself
        pass
    lib = ctypes.CDLL(path)
    lib.rdvio_hip_version.restype = ctypes.c_char_p
    lib.rdvio_hip_last_error.restype = ctypes.c_char_p
    lib.rdvio_hip_last_error.argtypes = [ctypes.c_void_p]
    lib.rdvio_hip_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    lib.rdvio_hip_ctx_destroy.argtypes = [ctypes.c_void_p]
    lib.rdvio_hip_ctx_destroy.restype = None
    lib.rdvio_hip_sync.argtypes = [ctypes.c_void_p]
    lib.rdvio_hip_ctx_set_lane_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.rdvio_hip_ctx_set_wait_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.rdvio_hip_lane_wait.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rdvio_hip_lane_sync.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.rdvio_hip_pyr_layout_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(PyrLayout)]
    lib.rdvio_hip_frame_step.argtypes = [ctypes.POINTER(FrameStep), ctypes.c_int]
    lib.rdvio_hip_run_sequences.argtypes = [ctypes.POINTER(FrameStep), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
    img_args = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                ctypes.c_double, ctypes.c_int, ctypes.c_int]
    lib.rdvio_hip_image_preprocess.argtypes = img_args
    lib.rdvio_hip_image_preprocess_dev.argtypes = img_args
    lib.rdvio_hip_image_download.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.rdvio_hip_image_release.argtypes = [ctypes.c_void_p, ctypes.c_int]
    trk = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
           ctypes.c_void_p]
    lib.rdvio_hip_track_keypoints.argtypes = trk
    lib.rdvio_hip_track_keypoints_dev.argtypes = trk
    lib.rdvio_hip_lk_flow.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
    lib.rdvio_hip_detect_keypoints.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                               ctypes.POINTER(ctypes.c_int)]
    lib.rdvio_hip_harris_response.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.rdvio_hip_preintegrate.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6 + [
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.rdvio_hip_reprojection_eval.argtypes = [ctypes.c_void_p, ctypes.POINTER(BaProblem)] + [ctypes.c_void_p] * 4
    lib.rdvio_hip_rotation_prior_eval.argtypes = [ctypes.c_void_p, ctypes.POINTER(BaProblem), ctypes.c_void_p, ctypes.c_void_p]
    lib.rdvio_hip_ba_solve.argtypes = [ctypes.c_void_p, ctypes.POINTER(BaProblem), ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.POINTER(BaSummary)]
    lib.rdvio_hip_ba_upload.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(BaProblem)]
    lib.rdvio_hip_ba_upload_chained.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(BaProblem), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rdvio_hip_preintegrate_estimator.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.rdvio_hip_preintegrate_estimator_begin.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int]
    lib.rdvio_hip_preintegrate_estimator_end.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rdvio_hip_debug_last_select_path.argtypes = [ctypes.c_void_p]
    lib.rdvio_hip_debug_last_select_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rdvio_hip_ba_solve_resident.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rdvio_hip_ba_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(BaSummary)]
    lib.rdvio_hip_marginalize.argtypes = [ctypes.c_void_p, ctypes.POINTER(MargProblem), ctypes.c_int] + \
        [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_int)]
    lib.rdvio_hip_marginalize_upload.argtypes = [ctypes.c_void_p, ctypes.POINTER(MargProblem)]
    lib.rdvio_hip_marginalize_resident.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.rdvio_hip_marginalize_fetch.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_int)]
    lib.rdvio_hip_preintegrate_dev.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    _LIB = lib
    return lib


def have_gpu():
    """True if a HIP device is usable (counts devices only; does not initialise a context)."""
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Context:
    """rdvio_hip_ctx: owns the device buffers; one per process/stream."""

    def __init__(self, max_width=752, max_height=480, max_features=1024, max_window=16, max_factors=16384,
                 device=0, stream=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        rc = self._lib.rdvio_hip_ctx_create(ctypes.byref(self._h), device, max_width, max_height, max_features,
                                            max_window, max_factors, stream)
        if rc != 0:
            msg = self._lib.rdvio_hip_last_error(self._h).decode() if self._h else "no HIP device / bad arguments"
            if self._h:
                self._lib.rdvio_hip_ctx_destroy(self._h)
                self._h = None
            raise RdvioError(f"rdvio_hip_ctx_create failed ({rc}): {msg}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rdvio_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, summary=None):
        if rc != 0:
            e = RdvioError(f"rdvio_hip error {rc}: {self._lib.rdvio_hip_last_error(self._h).decode()}")
            e.code, e.summary = rc, summary
            raise e

    def sync(self):
        self._check(self._lib.rdvio_hip_sync(self._h))

    def set_lane_stream(self, lane, stream=None):
        """give the solver (1) or marginalisation (2) lane a stream of its own (None: context-owned)"""
        self._check(self._lib.rdvio_hip_ctx_set_lane_stream(self._h, int(lane), stream))

    def lane_wait(self, lane, on_lane):
        self._check(self._lib.rdvio_hip_lane_wait(self._h, int(lane), int(on_lane)))

    def lane_sync(self, lane):
        self._check(self._lib.rdvio_hip_lane_sync(self._h, int(lane)))

    # ---------------------------------------------------------------- seam 2: estimation
    def preintegrate(self, segments, t_end, bg, ba, noise, jac=True, cov=True):
        """PreIntegrator::integrate for a batch of segments (list of n_i x 7 arrays). -> (nseg, 506)"""
        nseg = len(segments)
        off = np.zeros(nseg + 1, dtype=np.int32)
        for i, s in enumerate(segments):
            off[i + 1] = off[i] + len(s)
        imu = _f64(np.concatenate([np.asarray(s, dtype=np.float64).reshape(-1, 7) for s in segments], axis=0)) \
            if off[-1] > 0 else np.zeros((1, 7))
        t_end, bg, ba, noise = _f64(t_end).reshape(nseg), _f64(bg).reshape(nseg, 3), _f64(ba).reshape(nseg, 3), _f64(noise)
        out = np.zeros((nseg, PREINT_SIZE))
        self._check(self._lib.rdvio_hip_preintegrate(self._h, nseg, off.ctypes.data, imu.ctypes.data, t_end.ctypes.data,
                                                     bg.ctypes.data, ba.ctypes.data, noise.ctypes.data, int(jac),
                                                     int(cov), out.ctypes.data))
        return out

    def _ba_problem(self, pb):
        """problem dict -> rdvio_ba_problem (optional keys default to 'none')"""
        states = _f64(pb["states"]).reshape(-1, 16)
        n, nl = len(states), len(pb["inv_depth"])
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)  # noqa: E731
        k = dict(
            states=states, frame_fixed=u8(pb.get("frame_fixed", np.zeros(n))), extr=_f64(pb["extr"]),
            W=_f64(pb["sqrt_inv_cov"]), z_ref=_f64(pb["z_ref"]).reshape(-1, 3), inv_depth=_f64(pb["inv_depth"]),
            lm_fixed=u8(pb.get("lm_fixed", np.zeros(nl))), tgt=_i32(pb["tgt"]), ref=_i32(pb["ref"]), lm=_i32(pb["lm"]),
            tangent=_f64(pb["tangent"]).reshape(-1, 9), rot_tgt=_i32(pb.get("rot_tgt", [])),
            rot_ref=_i32(pb.get("rot_ref", [])), rot_zref=_f64(pb.get("rot_zref", np.zeros((0, 3)))),
            rot_tangent=_f64(pb.get("rot_tangent", np.zeros((0, 9)))), pre_i=_i32(pb.get("pre_i", [])),
            pre_j=_i32(pb.get("pre_j", [])), preint=_f64(pb.get("preint", np.zeros((0, PREINT_SIZE)))),
            prior_frames=_i32(pb.get("prior_frames", [])), lin=_f64(pb.get("lin", np.zeros((0, 16)))),
            S=_f64(pb.get("S", np.zeros((0, 0)))), f=_f64(pb.get("f", np.zeros(0))))
        c = BaProblem()
        c.n_frames, c.n_landmarks, c.n_factors = n, nl, len(k["tgt"])
        c.states, c.frame_fixed = k["states"].ctypes.data, k["frame_fixed"].ctypes.data
        c.extr, c.sqrt_inv_cov = k["extr"].ctypes.data, k["W"].ctypes.data
        c.z_ref, c.inv_depth, c.lm_fixed = k["z_ref"].ctypes.data, k["inv_depth"].ctypes.data, k["lm_fixed"].ctypes.data
        c.tgt, c.ref, c.lm, c.tangent = (k["tgt"].ctypes.data, k["ref"].ctypes.data, k["lm"].ctypes.data,
                                         k["tangent"].ctypes.data)
        c.n_rot = len(k["rot_tgt"])
        c.rot_tgt, c.rot_ref = k["rot_tgt"].ctypes.data, k["rot_ref"].ctypes.data
        c.rot_zref, c.rot_tangent = k["rot_zref"].ctypes.data, k["rot_tangent"].ctypes.data
        c.n_preint = len(k["pre_i"])
        c.pre_i, c.pre_j, c.preint = k["pre_i"].ctypes.data, k["pre_j"].ctypes.data, k["preint"].ctypes.data
        c.n_prior = len(k["prior_frames"])
        c.prior_frames, c.prior_lin = k["prior_frames"].ctypes.data, k["lin"].ctypes.data
        c.prior_S, c.prior_f = k["S"].ctypes.data, k["f"].ctypes.data
        return c, k

    def ba_solve(self, pb, max_iterations=30):
        """Solver::solve on a problem dict -> (states, inv_depth, BaSummary)"""
        c, keep = self._ba_problem(pb)
        states = np.zeros((c.n_frames, 16))
        invd = np.zeros(c.n_landmarks)
        sm = BaSummary()
        self._check(self._lib.rdvio_hip_ba_solve(self._h, ctypes.byref(c), int(max_iterations), states.ctypes.data,
                                                 invd.ctypes.data, ctypes.byref(sm)), sm)
        return states, invd, sm

    def ba_linearize(self, pb, lin_states=None, robust_loss=True):
        """rdvio_hip_ba_linearize: one linearisation with the solver's device routines -> dict of the pieces"""
        c, keep = self._ba_problem(pb)
        npre, D, nl = c.n_preint, 15 * c.n_prior, c.n_landmarks
        nfree = int((keep["frame_fixed"] != 1).sum())
        N = 15 * nfree
        out = dict(r_preint=np.zeros((npre, 15)), J_preint=np.zeros((npre, 2, 15, 15)), r_prior=np.zeros(D), J_prior=np.zeros((D, D)),
                   H=np.zeros((N, N)), g=np.zeros(N), lm_info=np.zeros(nl), lm_grad=np.zeros(nl), S_reduced=np.zeros((N, N)), c_reduced=np.zeros(N))
        lin = BaLinearization()
        for k, v in out.items():
            setattr(lin, k, v.ctypes.data if v.size else None)
        ls = _f64(lin_states).reshape(-1, 16) if lin_states is not None else None
        self._check(self._lib.rdvio_hip_ba_linearize(self._h, ctypes.byref(c), ls.ctypes.data_as(ctypes.c_void_p) if ls is not None else None,
                                                     1 if robust_loss else 0, ctypes.byref(lin)))
        assert lin.N == N
        return out

    def ba_upload(self, pb, slot=0):
        c, keep = self._ba_problem(pb)
        self._check(self._lib.rdvio_hip_ba_upload(self._h, int(slot), ctypes.byref(c)))
        self.sync()  # the source arrays may be freed by the caller after this returns
        if not hasattr(self, "_ba_shape"):
            self._ba_shape = {}
        self._ba_shape[int(slot)] = (c.n_frames, c.n_landmarks)

    def ba_upload_chained(self, pb, slot, from_slot, from_frame, to_frame):
        """rdvio_hip_ba_upload_chained: frame `to_frame` of this problem starts from the result of frame `from_frame` of the solve in
        `from_slot` (taken on the device, in stream order)."""
        c, keep = self._ba_problem(pb)
        self._check(self._lib.rdvio_hip_ba_upload_chained(self._h, int(slot), ctypes.byref(c), int(from_slot), int(from_frame), int(to_frame)))
        if not hasattr(self, "_ba_shape"):
            self._ba_shape = {}
        self._ba_shape[int(slot)] = (c.n_frames, c.n_landmarks)
        self._ba_keep = keep   # (the pinned blob was packed during the call; kept anyway until the next upload)

    def preintegrate_estimator(self, segments, t_end, bg, ba, noise, two_halves=False):
        """rdvio_hip_preintegrate_estimator (solver lane, staging of its own), in one call or as _begin / _end. -> (nseg, 506)"""
        nseg = len(segments)
        off = np.zeros(nseg + 1, dtype=np.int32)
        for i, s in enumerate(segments):
            off[i + 1] = off[i] + len(s)
        imu = _f64(np.concatenate([np.asarray(s, dtype=np.float64).reshape(-1, 7) for s in segments], axis=0))
        t_end, bg, ba, noise = _f64(t_end).reshape(nseg), _f64(bg).reshape(nseg, 3), _f64(ba).reshape(nseg, 3), _f64(noise)
        out = np.zeros((nseg, PREINT_SIZE))
        args = (self._h, nseg, off.ctypes.data, imu.ctypes.data, t_end.ctypes.data, bg.ctypes.data, ba.ctypes.data, noise.ctypes.data, 1, 1)
        if two_halves:
            self._check(self._lib.rdvio_hip_preintegrate_estimator_begin(*args))
            self._check(self._lib.rdvio_hip_preintegrate_estimator_end(self._h, out.ctypes.data))
        else:
            self._check(self._lib.rdvio_hip_preintegrate_estimator(*args, out.ctypes.data))
        return out

    def ba_solve_resident(self, max_iterations=30, slot=0):
        self._check(self._lib.rdvio_hip_ba_solve_resident(self._h, int(slot), int(max_iterations)))

    def ba_fetch(self, slot=0):
        n, nl = self._ba_shape[int(slot)]
        states, invd, sm = np.zeros((n, 16)), np.zeros(nl), BaSummary()
        self._check(self._lib.rdvio_hip_ba_fetch(self._h, int(slot), states.ctypes.data, invd.ctypes.data,
                                                 ctypes.byref(sm)), sm)
        return states, invd, sm

    def _marg_problem(self, states, extr, W, prior_frames, lin, S, f, preint01, tgt, ref, lm, tangent, z_ref, inv_depth):
        k = dict(states=_f64(states).reshape(-1, 16), extr=_f64(extr), W=_f64(W), pf=_i32(prior_frames), lin=_f64(lin),
                 S=_f64(S), f=_f64(f), pre=_f64(preint01) if preint01 is not None else None, tgt=_i32(tgt),
                 ref=_i32(ref), lm=_i32(lm), tangent=_f64(tangent), z_ref=_f64(z_ref), inv_depth=_f64(inv_depth))
        c = MargProblem()
        c.n_frames = len(k["states"])
        c.states, c.extr, c.sqrt_inv_cov = k["states"].ctypes.data, k["extr"].ctypes.data, k["W"].ctypes.data
        c.n_prior = len(k["pf"])
        c.prior_frames, c.prior_lin = k["pf"].ctypes.data, k["lin"].ctypes.data
        c.prior_S, c.prior_f = k["S"].ctypes.data, k["f"].ctypes.data
        c.preint01 = k["pre"].ctypes.data if k["pre"] is not None else None
        c.n_landmarks = len(k["inv_depth"])
        c.z_ref, c.inv_depth = k["z_ref"].ctypes.data, k["inv_depth"].ctypes.data
        c.n_factors = len(k["tgt"])
        c.tgt, c.ref, c.lm, c.tangent = k["tgt"].ctypes.data, k["ref"].ctypes.data, k["lm"].ctypes.data, k["tangent"].ctypes.data
        return c, k

    def marginalize(self, states, extr, W, prior_frames, lin, S, f, preint01, tgt, ref, lm, tangent, z_ref, inv_depth,
                    force_eigen=False):
        """MarginalizationFactor::marginalize(0) -> (S, f, lin, Lambda, eta, used_fast_path)"""
        c, keep = self._marg_problem(states, extr, W, prior_frames, lin, S, f, preint01, tgt, ref, lm, tangent, z_ref,
                                     inv_depth)
        R = 15 * (c.n_frames - 1)
        S_out, f_out, lin_out = np.zeros((R, R)), np.zeros(R), np.zeros((c.n_frames - 1, 16))
        Lam, eta, fast = np.zeros((R, R)), np.zeros(R), ctypes.c_int(0)
        self._check(self._lib.rdvio_hip_marginalize(self._h, ctypes.byref(c), int(force_eigen), S_out.ctypes.data,
                                                    f_out.ctypes.data, lin_out.ctypes.data, Lam.ctypes.data,
                                                    eta.ctypes.data, ctypes.byref(fast)))
        return S_out, f_out, lin_out, Lam, eta, bool(fast.value)

    def marginalize_upload(self, *args):
        c, keep = self._marg_problem(*args)
        self._check(self._lib.rdvio_hip_marginalize_upload(self._h, ctypes.byref(c)))
        self.sync()

    def marginalize_resident(self, force_eigen=False):
        self._check(self._lib.rdvio_hip_marginalize_resident(self._h, int(force_eigen)))

    def reprojection_eval(self, pb, jac=True):
        """CeresReprojectionErrorFactor::Evaluate over all factors of a BA problem dict."""
        c, keep = self._ba_problem(pb)
        n = c.n_factors
        r = np.zeros((n, 2))
        Jt = np.zeros((n, 2, 6)) if jac else None
        Jr = np.zeros((n, 2, 6)) if jac else None
        Jd = np.zeros((n, 2)) if jac else None
        self._check(self._lib.rdvio_hip_reprojection_eval(
            self._h, ctypes.byref(c), r.ctypes.data, Jt.ctypes.data if jac else None,
            Jr.ctypes.data if jac else None, Jd.ctypes.data if jac else None))
        return r, Jt, Jr, Jd


def _rotation_prior_eval(self, pb, jac=True):
    """CeresRotationPriorFactor::Evaluate over the rotation priors of a BA problem dict -> (r (n,2), J (n,2,3) or None)"""
    c, keep = self._ba_problem(pb)
    n = c.n_rot
    r = np.zeros((n, 2))
    J = np.zeros((n, 2, 3)) if jac else None
    self._check(self._lib.rdvio_hip_rotation_prior_eval(self._h, ctypes.byref(c), r.ctypes.data, J.ctypes.data if jac else None))
    return r, J


Context.rotation_prior_eval = _rotation_prior_eval


class HipImage:
    """Mirror of rdvio::Image (types.h:153-177) / OpenCvImage on a context image slot."""

    def __init__(self, ctx, slot, gray):
        self.ctx = ctx
        self.slot = int(slot)
        self.image = np.ascontiguousarray(gray, dtype=np.uint8)
        if self.image.ndim != 2:
            raise RdvioError("HipImage expects a single-channel u8 image")
        self.L = None

    def width(self):
        return self.image.shape[1]

    def height(self):
        return self.image.shape[0]

    def level_num(self):
        return 3

    def preprocess(self, clip_limit=6.0, width=8, height=8):
        """Image::preprocess(clipLimit, width, height) -- opencv_image.cpp:156-161"""
        h, w = self.image.shape
        c = self.ctx
        c._check(c._lib.rdvio_hip_image_preprocess(c._h, self.slot, self.image.ctypes.data, w, h, w, float(clip_limit),
                                                   int(width), int(height)))
        self.L = PyrLayout()
        c._lib.rdvio_hip_pyr_layout_init(w, h, MAX_LEVELS - 1, ctypes.byref(self.L))

    def download(self):
        """(pyr_img u8 arena, pyr_deriv int16 arena) -- test helper"""
        c = self.ctx
        pi = np.zeros(self.L.img_bytes, dtype=np.uint8)
        pd = np.zeros(self.L.deriv_elems, dtype=np.int16)
        c._check(c._lib.rdvio_hip_image_download(c._h, self.slot, pi.ctypes.data, pd.ctypes.data))
        return pi, pd

    def track_keypoints(self, next_image, curr_keypoints, next_keypoints=None):
        """Image::track_keypoints(next_image, curr, next_inout, status) -- opencv_image.cpp:75-154.
        Returns (next_keypoints (n,2) double, status (n,) u8)."""
        c = self.ctx
        curr = _f64(curr_keypoints).reshape(-1, 2)
        n = len(curr)
        has_guess = next_keypoints is not None and len(next_keypoints) > 0
        nxt = _f64(next_keypoints).reshape(-1, 2).copy() if has_guess else np.zeros((n, 2))
        st = np.zeros(n, dtype=np.uint8)
        c._check(c._lib.rdvio_hip_track_keypoints(c._h, self.slot, next_image.slot, n, curr.ctypes.data,
                                                  nxt.ctypes.data, int(has_guess), st.ctypes.data))
        return nxt, st

    def lk_flow(self, next_image, prev_xy, next_xy, max_iter=30, eps=0.01):
        c = self.ctx
        prev = np.ascontiguousarray(prev_xy, dtype=np.float32).reshape(-1, 2)
        nxt = np.ascontiguousarray(next_xy, dtype=np.float32).reshape(-1, 2).copy()
        st = np.zeros(len(prev), dtype=np.uint8)
        c._check(c._lib.rdvio_hip_lk_flow(c._h, self.slot, next_image.slot, len(prev), prev.ctypes.data,
                                          nxt.ctypes.data, st.ctypes.data, int(max_iter), float(eps)))
        return nxt, st

    def harris_response(self):
        c = self.ctx
        h, w = self.image.shape
        out = np.zeros((h, w), dtype=np.float32)
        c._check(c._lib.rdvio_hip_harris_response(c._h, self.slot, out.ctypes.data))
        return out

    def detect_keypoints(self, keypoints, max_points=1000, keypoint_distance=10.0):
        """Image::detect_keypoints(keypoints_inout, max_points, distance) -- opencv_image.cpp:38-73"""
        c = self.ctx
        existing = _f64(keypoints).reshape(-1, 2)
        cap = len(existing) + int(max_points)
        buf = np.zeros((cap, 2))
        buf[:len(existing)] = existing
        n_out = ctypes.c_int(0)
        c._check(c._lib.rdvio_hip_detect_keypoints(c._h, self.slot, buf.ctypes.data, len(existing), cap,
                                                   int(max_points), float(keypoint_distance), ctypes.byref(n_out)))
        return buf[:n_out.value].copy()

    def release_image_buffer(self):
        c = self.ctx
        c._check(c._lib.rdvio_hip_image_release(c._h, self.slot))
