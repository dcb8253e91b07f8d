"""Pins for the CPU oracle's estimation math (the reference has no
fixtures, so the restatement is pinned by finite differences and algebraic identities)."""
import numpy as np
import pytest

from rd_vio_amd import synth


def _perturb_state(s, d15):
    """right-multiplicative quaternion update + additive rest (quaternion_parameterization.h:11-17)."""
    o = s.copy()
    q = synth.q_mul(s[0:4], synth.q_exp(d15[0:3]))
    o[0:4] = q / np.linalg.norm(q)
    o[4:16] += d15[3:15]
    return o


def test_expmap_logmap_roundtrip(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        w = rng.normal(0, 1.0, 3)
        if np.linalg.norm(w) > 3.0:
            continue
        assert np.allclose(oracle.logmap(oracle.expmap(w)), w, atol=1e-12)
    assert np.allclose(oracle.expmap(np.zeros(3)), [0, 0, 0, 1])
    assert np.allclose(oracle.logmap(np.array([0, 0, 0, 1.0])), 0)
    # w < 0 branch of Eigen's AngleAxis(q): same rotation, angle in [0, pi]
    q = oracle.expmap(np.array([0.3, -0.2, 0.1]))
    assert np.allclose(oracle.logmap(-q), oracle.logmap(q), atol=1e-12)


def test_right_jacobian_fd(oracle):
    rng = np.random.default_rng(1)
    for scale in (1e-9, 1e-3, 0.5):
        w = rng.normal(0, scale, 3)
        J = oracle.right_jacobian(w)
        h = 1e-6
        Jn = np.zeros((3, 3))
        for k in range(3):
            d = np.zeros(3)
            d[k] = h
            qp = synth.q_mul(synth.q_conj(oracle.expmap(w)), oracle.expmap(w + d))
            qm = synth.q_mul(synth.q_conj(oracle.expmap(w)), oracle.expmap(w - d))
            Jn[:, k] = (oracle.logmap(qp) - oracle.logmap(qm)) / (2 * h)
        assert np.allclose(J, Jn, atol=1e-6)


def test_tangent_frame_orthonormal(oracle):
    rng = np.random.default_rng(2)
    for _ in range(20):
        z = rng.normal(size=3)
        z /= np.linalg.norm(z)
        T = oracle.tangent_frame(z)
        assert np.allclose(T.T @ T, np.eye(3), atol=1e-12)
        assert np.allclose(T[:, 2], z)
        assert np.allclose(T, synth.tangent_frame(z))


def test_reprojection_jacobian_fd(oracle):
    pb = synth.make_ba_problem(n_frames=5, n_landmarks=40, seed=3)
    r, Jt, Jr, Jd = oracle.reprojection_eval(pb["tgt"], pb["ref"], pb["lm"], pb["tangent"], pb["z_ref"],
                                             pb["inv_depth"], pb["states"], pb["extr"], pb["sqrt_inv_cov"])
    assert np.isfinite(r).all() and len(r) > 50
    h = 1e-6
    for k in range(0, len(r), 7):
        t, f, l = pb["tgt"][k], pb["ref"][k], pb["lm"][k]

        def res(states, inv_depth):
            return oracle.reprojection_eval(pb["tgt"][k:k + 1], pb["ref"][k:k + 1], pb["lm"][k:k + 1],
                                            pb["tangent"][k:k + 1], pb["z_ref"], inv_depth, states, pb["extr"],
                                            pb["sqrt_inv_cov"], jac=False)[0][0]

        for frame, J in ((t, Jt[k]), (f, Jr[k])):
            for c in range(6):
                d = np.zeros(15)
                d[c] = h
                sp = pb["states"].copy()
                sm = pb["states"].copy()
                sp[frame] = _perturb_state(sp[frame], d)
                sm[frame] = _perturb_state(sm[frame], -d)
                num = (res(sp, pb["inv_depth"]) - res(sm, pb["inv_depth"])) / (2 * h)
                assert np.allclose(J[:, c], num, rtol=1e-5, atol=1e-4 * max(1.0, np.abs(num).max()))
        ip, im = pb["inv_depth"].copy(), pb["inv_depth"].copy()
        ip[l] += h
        im[l] -= h
        num = (res(pb["states"], ip) - res(pb["states"], im)) / (2 * h)
        assert np.allclose(Jd[k], num, rtol=1e-5, atol=1e-3)


def test_reprojection_zero_residual_at_truth(oracle):
    pb = synth.make_ba_problem(n_frames=4, n_landmarks=30, seed=4, pix_noise=0.0, state_noise=False)
    r, *_ = oracle.reprojection_eval(pb["tgt"], pb["ref"], pb["lm"], pb["tangent"], pb["z_ref"], pb["inv_depth"],
                                     pb["states"], pb["extr"], pb["sqrt_inv_cov"])
    assert np.abs(r).max() < 1e-8


def _make_preint(oracle, t0, t1, bg, ba, seed=5, noise=True):
    rng = np.random.default_rng(seed)
    imu = synth.make_imu_segment(t0, t1, rng=rng, noise=noise)
    return oracle.preintegrate(imu, t1, bg, ba, synth.EUROC_NOISE), imu


def test_preintegration_constant_motion_closed_form(oracle):
    # constant body rate w about z and constant specific force a, zero bias: closed form for dq; dt sums
    n, dt = 20, 0.005
    w = np.array([0.0, 0.0, 0.4])
    a = np.array([0.1, -0.2, 9.8])
    imu = np.array([[i * dt, *w, *a] for i in range(n)])
    pre = oracle.preintegrate(imu, n * dt, np.zeros(3), np.zeros(3), synth.EUROC_NOISE)
    assert abs(pre[oracle.PREINT_T] - n * dt) < 1e-15
    assert np.allclose(pre[oracle.PREINT_Q:oracle.PREINT_Q + 4], synth.q_exp(w * n * dt), atol=1e-12)
    # cov symmetric PSD and sqrt_inv_cov^T sqrt_inv_cov == cov^-1 (preintegrator.cpp:97-100)
    cov = pre[oracle.PREINT_COV:oracle.PREINT_COV + 225].reshape(15, 15)
    U = pre[oracle.PREINT_SIC:oracle.PREINT_SIC + 225].reshape(15, 15)
    assert np.allclose(cov, cov.T, rtol=1e-9, atol=1e-20)
    assert np.allclose(U, np.triu(U))
    assert np.allclose(U.T @ U @ cov, np.eye(15), atol=1e-6)


def test_preintegration_matches_trajectory(oracle):
    # noise-free IMU from the analytic trajectory: predict() must land on the true next state
    t0, t1 = 1.0, 1.25
    q0, p0 = synth.traj_pose(t0)
    s0 = np.concatenate([q0, p0, synth.traj_vel(t0), np.zeros(6)])
    rng = np.random.default_rng(0)
    imu = synth.make_imu_segment(t0, t1, rate=2000.0, rng=rng, noise=False)
    pre = oracle.preintegrate(imu, t1, np.zeros(3), np.zeros(3), synth.EUROC_NOISE)
    s1 = oracle.preint_predict(pre, s0)
    q1, p1 = synth.traj_pose(t1)
    assert np.allclose(s1[4:7], p1, atol=2e-4)
    assert np.allclose(s1[7:10], synth.traj_vel(t1), atol=2e-3)
    assert abs(abs(np.dot(s1[0:4], q1)) - 1) < 1e-6


def test_preintegration_bias_jacobians_fd(oracle):
    rng = np.random.default_rng(6)
    imu = synth.make_imu_segment(1.0, 1.1, rng=rng)
    bg, ba = np.array([1e-3, -2e-3, 5e-4]), np.array([0.02, -0.01, 0.03])
    pre = oracle.preintegrate(imu, 1.1, bg, ba, synth.EUROC_NOISE)
    jac = pre[oracle.PREINT_JAC:].reshape(5, 3, 3)  # dq_dbg dp_dbg dp_dba dv_dbg dv_dba
    h = 1e-6
    for k in range(3):
        d = np.zeros(3)
        d[k] = h
        pp = oracle.preintegrate(imu, 1.1, bg + d, ba, synth.EUROC_NOISE)
        pm = oracle.preintegrate(imu, 1.1, bg - d, ba, synth.EUROC_NOISE)
        dq = oracle.logmap(synth.q_mul(synth.q_conj(pm[1:5]), pp[1:5])) / (2 * h)
        assert np.allclose(jac[0][:, k], dq, atol=1e-5)
        assert np.allclose(jac[1][:, k], (pp[5:8] - pm[5:8]) / (2 * h), atol=1e-5)
        assert np.allclose(jac[3][:, k], (pp[8:11] - pm[8:11]) / (2 * h), atol=1e-5)
        pp = oracle.preintegrate(imu, 1.1, bg, ba + d, synth.EUROC_NOISE)
        pm = oracle.preintegrate(imu, 1.1, bg, ba - d, synth.EUROC_NOISE)
        assert np.allclose(jac[2][:, k], (pp[5:8] - pm[5:8]) / (2 * h), atol=1e-5)
        assert np.allclose(jac[4][:, k], (pp[8:11] - pm[8:11]) / (2 * h), atol=1e-5)


def test_preintegration_factor_jacobian_fd(oracle):
    rng = np.random.default_rng(7)
    t0, t1 = 1.0, 1.25
    bias = np.array([1e-3, -2e-3, 5e-4, 0.02, -0.01, 0.03])
    pre, _ = _make_preint(oracle, t0, t1, bias[:3], bias[3:])
    q0, p0 = synth.traj_pose(t0)
    q1, p1 = synth.traj_pose(t1)
    si = np.concatenate([q0, p0, synth.traj_vel(t0), bias]) 
    sj = np.concatenate([q1, p1, synth.traj_vel(t1), bias + rng.normal(0, 1e-4, 6)])
    si = _perturb_state(si, rng.normal(0, 1e-3, 15))
    sj = _perturb_state(sj, rng.normal(0, 1e-3, 15))
    extr = synth.EUROC_EXTR.copy()
    extr[7:11] = synth.q_exp(np.array([0.1, -0.05, 0.02]))  # non-trivial imu extrinsics exercise every block
    extr[11:14] = [0.01, -0.02, 0.03]
    r, Ji, Jj = oracle.preintegration_eval(si, sj, pre, bias, extr)
    U = pre[oracle.PREINT_SIC:oracle.PREINT_SIC + 225].reshape(15, 15)
    h = 1e-7
    for which, J in ((0, Ji), (1, Jj)):
        for c in range(15):
            d = np.zeros(15)
            d[c] = h
            a = [si, sj]
            b = [si, sj]
            a[which] = _perturb_state(a[which], d)
            b[which] = _perturb_state(b[which], -d)
            rp = oracle.preintegration_eval(a[0], a[1], pre, bias, extr, jac=False)[0]
            rm = oracle.preintegration_eval(b[0], b[1], pre, bias, extr, jac=False)[0]
            num = (rp - rm) / (2 * h)
            scale = np.abs(U).sum(axis=1) + 1.0
            assert np.allclose(J[:, c] / scale, num / scale, atol=2e-4), (which, c)
    # whitening: r == U e  => residual small at the (noisy-IMU) truth compared to a perturbed state
    assert np.isfinite(r).all()


def test_marginalization_eval_fd_and_initial_prior(oracle):
    rng = np.random.default_rng(8)
    n = 3
    D = 15 * n
    lin = synth.make_ba_problem(n_frames=n, n_landmarks=5, seed=8)["states_true"]
    states = np.stack([_perturb_state(lin[i], rng.normal(0, 1e-2, 15)) for i in range(n)])
    A = rng.normal(size=(D, D))
    S = A
    f = rng.normal(size=D)
    r, J = oracle.marginalization_eval(states, lin, S, f)
    h = 1e-6
    for i in range(n):
        for c in range(15):
            d = np.zeros(15)
            d[c] = h
            sp, sm = states.copy(), states.copy()
            sp[i] = _perturb_state(sp[i], d)
            sm[i] = _perturb_state(sm[i], -d)
            num = (oracle.marginalization_eval(sp, lin, S, f, jac=False)[0]
                   - oracle.marginalization_eval(sm, lin, S, f, jac=False)[0]) / (2 * h)
            assert np.allclose(J[:, 15 * i + c], num, atol=1e-5)
    # at the linearisation point r == f
    r0, _ = oracle.marginalization_eval(lin, lin, S, f)
    assert np.allclose(r0, f)


def _marg_inputs(oracle, n_frames=5, n_landmarks=40, seed=9):
    pb = synth.make_ba_problem(n_frames=n_frames, n_landmarks=n_landmarks, seed=seed)
    rng = np.random.default_rng(seed)
    npf = n_frames - 1
    D = 15 * npf
    A = rng.normal(size=(D, D)) * 3.0
    S = np.linalg.qr(A)[1]  # some full-rank sqrt information
    f = rng.normal(size=D)
    lin = np.stack([_perturb_state(pb["states"][i], rng.normal(0, 1e-3, 15)) for i in range(npf)])
    bias = pb["states"][0, 10:16]
    imu = synth.make_imu_segment(1.0, 1.25, rng=rng)
    pre = oracle.preintegrate(imu, 1.25, bias[:3], bias[3:], synth.EUROC_NOISE)
    # factors of tracks observed by the victim (frame 0): anchor == 0 or a target == 0
    seen = set(pb["lm"][(pb["tgt"] == 0) | (pb["ref"] == 0)].tolist())
    keep = np.array([l in seen for l in pb["lm"]])
    return pb, np.arange(npf, dtype=np.int32), lin, S, f, pre, keep


def test_marginalize_matches_dense_schur(oracle):
    pb, pf, lin, S, f, pre, keep = _marg_inputs(oracle)
    tgt, ref, lm, tan = pb["tgt"][keep], pb["ref"][keep], pb["lm"][keep], pb["tangent"][keep]
    S2, f2, lin2, Lam, eta = oracle.marginalize(pb["states"], pb["extr"], pb["sqrt_inv_cov"], pf, lin, S, f, pre,
                                                tgt, ref, lm, tan, pb["z_ref"], pb["inv_depth"])
    n = len(pb["states"])
    L = len(pb["inv_depth"])
    # dense normal equations over [frames(15 n) | landmarks(L)], then eliminate landmarks + frame 0 at once
    N = 15 * n + L
    H = np.zeros((N, N))
    b = np.zeros(N)
    rP, JP = oracle.marginalization_eval(pb["states"][pf], lin, S, f)
    Jfull = np.zeros((len(rP), N))
    for i, fi in enumerate(pf):
        Jfull[:, 15 * fi:15 * fi + 15] = JP[:, 15 * i:15 * i + 15]
    H += Jfull.T @ Jfull
    b += Jfull.T @ rP
    rI, Ji, Jj = oracle.preintegration_eval(pb["states"][0], pb["states"][1], pre, pb["states"][0, 10:16], pb["extr"])
    Jfull = np.zeros((15, N))
    Jfull[:, 0:15] = Ji
    Jfull[:, 15:30] = Jj
    H += Jfull.T @ Jfull
    b += Jfull.T @ rI
    r, Jt, Jr, Jd = oracle.reprojection_eval(tgt, ref, lm, tan, pb["z_ref"], pb["inv_depth"], pb["states"],
                                             pb["extr"], pb["sqrt_inv_cov"])
    for k in range(len(r)):
        Jfull = np.zeros((2, N))
        Jfull[:, 15 * tgt[k]:15 * tgt[k] + 6] = Jt[k]
        Jfull[:, 15 * ref[k]:15 * ref[k] + 6] = Jr[k]
        Jfull[:, 15 * n + lm[k]] = Jd[k]
        H += Jfull.T @ Jfull
        b += Jfull.T @ r[k]
    used = np.unique(lm)
    elim = np.concatenate([np.arange(0, 15), 15 * n + used])
    keepi = np.arange(15, 15 * n)
    Hmm = H[np.ix_(elim, elim)]
    Hrm = H[np.ix_(keepi, elim)]
    Lam_ref = H[np.ix_(keepi, keepi)] - Hrm @ np.linalg.solve(Hmm, Hrm.T)
    eta_ref = b[keepi] - Hrm @ np.linalg.solve(Hmm, b[elim])
    scale = np.abs(Lam_ref).max()
    assert np.allclose(Lam, Lam_ref, atol=1e-9 * scale)
    assert np.allclose(eta, eta_ref, atol=1e-9 * np.abs(eta_ref).max())
    # sqrt form: S^T S == Lambda on the retained eigenspace, S^T f == eta there
    assert np.allclose(S2.T @ S2, Lam, atol=1e-8 * scale)
    assert np.allclose(S2.T @ f2, eta, atol=1e-6 * np.abs(eta).max())
    assert np.allclose(lin2, pb["states"][1:])


def test_sym_eig_via_marginalize_threshold(oracle):
    # rank-deficient information: eigenvalues <= 1e-8 are clamped to zero (marginalization_factor.h:445-450)
    pb, pf, lin, S, f, pre, keep = _marg_inputs(oracle, n_frames=3, n_landmarks=10, seed=11)
    S0 = np.zeros_like(S)
    S0[:6, :6] = 1e15 * np.eye(6)  # MarginalizationFactor ctor: pins pose of frame 0 (base:27-31)
    f0 = np.zeros_like(f)
    out = oracle.marginalize(pb["states"], pb["extr"], pb["sqrt_inv_cov"], pf, pb["states"][pf], S0, f0, pre,
                             pb["tgt"][keep], pb["ref"][keep], pb["lm"][keep], pb["tangent"][keep], pb["z_ref"],
                             pb["inv_depth"])
    S2, f2, _, Lam, eta = out
    ev = np.linalg.eigvalsh(Lam)
    ev2 = np.linalg.eigvalsh(S2.T @ S2)
    assert np.isfinite(S2).all() and np.isfinite(f2).all()
    assert np.allclose(np.where(ev > 1e-8, ev, 0), ev2, rtol=1e-6, atol=1e-6 * ev.max())


def test_rotation_prior_matches_numpy_restatement_and_fd(oracle):
    """Row A10, CeresRotationPriorFactor::Evaluate (ceres/rotation_factor.h:22-58): the C oracle against an independent
    numpy restatement of the same lines (incl. the translation added to a bearing, :34) and its Jacobian against central
    differences of the right-multiplicative update q <- q exp(d) (quaternion_parameterization.h:11-17)."""
    pb = synth.make_window_problem(6, 40, 31, preintegrate=lambda imu, t, bg, ba: oracle.preintegrate(imu, t, bg, ba, synth.EUROC_NOISE))
    synth.add_rotation_priors(pb, 25)
    assert len(pb["rot_tgt"]) == 25
    ex, W = pb["extr"], pb["sqrt_inv_cov"]
    Rcs, pcs = synth.q_to_mat(ex[0:4]), ex[4:7]

    def numpy_r(q_t, q_r, z, T):
        z_rc = Rcs @ z + pcs
        z_tc = synth.q_to_mat(q_t).T @ (synth.q_to_mat(q_r) @ z_rc)
        z_t = Rcs.T @ (z_tc - pcs)
        u = T.T @ z_t
        return W @ (u[:2] / u[2])

    h = 1e-6
    for k in range(len(pb["rot_tgt"])):
        q_t, q_r = pb["states"][pb["rot_tgt"][k], :4], pb["states"][pb["rot_ref"][k], :4]
        z, T = pb["rot_zref"][k], pb["rot_tangent"][k].reshape(3, 3)
        r, J = oracle.rotation_prior_eval(q_t, q_r, z, T, ex, W)
        assert np.allclose(r, numpy_r(q_t, q_r, z, T), rtol=1e-12, atol=1e-10)
        for c in range(3):
            d = np.zeros(3)
            d[c] = h
            qp = synth.q_mul(q_t, synth.q_exp(d))
            qm = synth.q_mul(q_t, synth.q_exp(-d))
            num = (numpy_r(qp / np.linalg.norm(qp), q_r, z, T) - numpy_r(qm / np.linalg.norm(qm), q_r, z, T)) / (2 * h)
            assert np.allclose(J[:, c], num, rtol=1e-5, atol=1e-4 * max(1.0, np.abs(num).max()))


def test_rotation_priors_enter_the_solve(oracle):
    """The oracle's solver uses the rotation priors (solver.cpp:134-141: CauchyLoss(1.0)): the initial cost grows by
    exactly the sum of 0.5 log(1 + |r|^2) over them, and they pull the free target frame's orientation."""
    pre = lambda imu, t, bg, ba: oracle.preintegrate(imu, t, bg, ba, synth.EUROC_NOISE)  # noqa: E731
    pb = synth.make_window_problem(6, 40, 32, preintegrate=pre)
    base = oracle.ba_solve(pb, 0)[2].initial_cost
    synth.add_rotation_priors(pb, 30)
    sm = oracle.ba_solve(pb, 0)[2]
    extra = 0.0
    for k in range(30):
        r, _ = oracle.rotation_prior_eval(pb["states"][pb["rot_tgt"][k], :4], pb["states"][pb["rot_ref"][k], :4], pb["rot_zref"][k],
                                          pb["rot_tangent"][k].reshape(3, 3), pb["extr"], pb["sqrt_inv_cov"])
        extra += 0.5 * np.log1p(r @ r)
    assert abs(sm.initial_cost - (base + extra)) <= 1e-9 * sm.initial_cost
