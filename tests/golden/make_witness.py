#!/usr/bin/env python3
"""Independent witness of the estimation rows' RESIDUALS ("a second witness"): a numpy restatement of

  A7   PreIntegrator::increment / integrate / compute_sqrt_inv_cov   /root/reference/src/rdvio_estimation/src/preintegrator.cpp:22-100
  A8   CeresReprojectionErrorFactor::Evaluate (residual)             .../ceres/reprojection_factor.h:16-50 (+ whitening :86)
  A11  CeresPreIntegrationErrorFactor::Evaluate (residual)           .../ceres/preintegration_factor.h:19-67, 155
  A12  CeresMarginalizationFactor::Evaluate (residual)               .../ceres/marginalization_factor.h:27-45, 68-69

written straight from those lines by a DIFFERENT ROUTE than oracle/*.c and the HIP kernels: rotations are 3 x 3 matrices
(Rodrigues' formula for exp, the matrix logarithm through the antisymmetric part for log), none of the oracle's helpers is
used and nothing is shared with it.  Finite differences (tests/test_oracle_estimation.py) prove that the oracle's Jacobians
match ITS residuals; this file pins the residuals themselves against a second transcription of the reference's formulas.

    python tests/golden/make_witness.py      -> tests/golden/witness_vectors.npz  (inputs + expected outputs, seeded)

The reference cannot be run here (C++17 + Eigen + Ceres, none installed), so this is a second reading, not the reference's own
output: parity stays "unpinned" in the sense of the task statement; what it removes is the single-transcriber risk.
"""
import os

import numpy as np

GRAVITY = np.array([0.0, 0.0, -9.80665])   # types.h:26


# ------------------------------------------------------------------ rotations as matrices
def hat(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def exp_so3(w):
    """Rodrigues: exp([w]x)"""
    th = np.linalg.norm(w)
    K = hat(w)
    if th < 1e-12:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1.0 - np.cos(th)) / th ** 2 * K @ K


def log_so3(R):
    """rotation vector of R through the antisymmetric part (angle from the trace and the norm of that part)"""
    a = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = np.linalg.norm(a), 0.5 * (np.trace(R) - 1.0)
    th = np.arctan2(s, c)
    if s < 1e-12:
        return a.copy()   # (angles near pi are not generated below)
    return a * (th / s)


def R_of_quat(q):
    """q = (x, y, z, w), Eigen's coefficient order (solver.cpp:90-91)"""
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_of_R(R):
    w = 0.5 * np.sqrt(max(1.0 + np.trace(R), 0.0))
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (4.0 * w)
    return np.array([v[0], v[1], v[2], w])


def right_jacobian(w):
    """lie_algebra.cpp:5-45 (the Taylor guards are irrelevant at the angles used here)"""
    th = np.linalg.norm(w)
    K = hat(w)
    if th < 1e-6:
        return np.eye(3) - 0.5 * K + K @ K / 6.0
    return np.eye(3) - (1.0 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K


# ------------------------------------------------------------------ A7
def preintegrate(imu, t_end, bg, ba, noise):
    """imu: n x 7 (t, gyro, acc); noise: cov_w cov_a cov_bg cov_ba (4 x 9).  Returns the 506-double record layout of
    include/rdvio_hip.h: t, q(4), p(3), v(3), cov(225), sqrt_inv_cov(225), dq_dbg dp_dbg dp_dba dv_dbg dv_dba (5 x 9)."""
    cov_w, cov_a, cov_bg, cov_ba = [noise[9 * k:9 * k + 9].reshape(3, 3) for k in range(4)]
    T, R, p, v = 0.0, np.eye(3), np.zeros(3), np.zeros(3)
    cov = np.zeros((15, 15))
    dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba = (np.zeros((3, 3)) for _ in range(5))
    n = len(imu)
    for i in range(n):
        dt = (imu[i + 1, 0] if i + 1 < n else t_end) - imu[i, 0]
        w, a = imu[i, 1:4] - bg, imu[i, 4:7] - ba
        E = exp_so3(w * dt)
        A = np.eye(9)                       # error-state order theta(0:3) p(3:6) v(6:9)  (state.h:11-18)
        A[0:3, 0:3] = E.T
        A[6:9, 0:3] = -dt * R @ hat(a)
        A[3:6, 0:3] = -0.5 * dt * dt * R @ hat(a)
        A[3:6, 6:9] = dt * np.eye(3)
        B = np.zeros((9, 6))
        B[0:3, 0:3] = dt * right_jacobian(w * dt)
        B[6:9, 3:6] = dt * R
        B[3:6, 3:6] = 0.5 * dt * dt * R
        Q = np.zeros((6, 6))
        inv_dt = 1.0 / max(dt, 1.0e-7)
        Q[0:3, 0:3] = cov_w * inv_dt
        Q[3:6, 3:6] = cov_a * inv_dt
        cov[0:9, 0:9] = A @ cov[0:9, 0:9] @ A.T + B @ Q @ B.T
        cov[9:12, 9:12] += cov_bg * dt
        cov[12:15, 12:15] += cov_ba * dt
        dp_dbg = dp_dbg + dt * dv_dbg - 0.5 * dt * dt * R @ hat(a) @ dq_dbg
        dp_dba = dp_dba + dt * dv_dba - 0.5 * dt * dt * R
        dv_dbg = dv_dbg - dt * R @ hat(a) @ dq_dbg
        dv_dba = dv_dba - dt * R
        dq_dbg = E.T @ dq_dbg - dt * right_jacobian(w * dt)
        T += dt
        p = p + dt * v + 0.5 * dt * dt * (R @ a)
        v = v + dt * (R @ a)
        R = R @ E
    L = np.linalg.cholesky(np.linalg.inv(cov))   # LLT(cov^-1).matrixL(); the record keeps its transpose (preintegrator.cpp:97-100)
    return np.concatenate([[T], quat_of_R(R), p, v, cov.ravel(), L.T.ravel(), dq_dbg.ravel(), dp_dbg.ravel(), dp_dba.ravel(), dv_dbg.ravel(), dv_dba.ravel()])


# ------------------------------------------------------------------ A8
def tangent_frame(z):
    """[b1 b2 z] (lie_algebra.cpp:47-56; reprojection_factor.h:16-22)"""
    d = int(np.argmax(np.abs(z)))   # (first maximum, like the loop's strict >)
    e = np.zeros(3)
    e[(d + 1) % 3] = 1.0
    b1 = np.cross(z, e)
    b1 /= np.linalg.norm(b1)
    b2 = np.cross(z, b1)
    b2 /= np.linalg.norm(b2)
    return np.stack([b1, b2, z], axis=1)


def reprojection_residual(state_tgt, state_ref, z_ref, inv_depth, T, extr, W):
    """state: q(4) p(3) ...; extr: camera q_cs(4) p_cs(3), imu q_cs p_cs; W = sqrt_inv_cov 2 x 2"""
    Rcs, pcs = R_of_quat(extr[0:4]), extr[4:7]
    Rt, pt = R_of_quat(state_tgt[0:4]), state_tgt[4:7]
    Rr, pr = R_of_quat(state_ref[0:4]), state_ref[4:7]
    y_ref = z_ref / inv_depth
    x = Rr @ (Rcs @ y_ref + pcs) + pr
    y_t = Rcs.T @ (Rt.T @ (x - pt) - pcs)
    u = T.T @ y_t
    return W @ (u[0:2] / u[2])


# ------------------------------------------------------------------ A11
def preintegration_residual(state_i, state_j, rec, bias_lin, extr):
    """rec: the 506-double record; bias_lin = (bg_i0, ba_i0), the biases the record was integrated about"""
    Ris, pis = R_of_quat(extr[7:11]), extr[11:14]       # imu extrinsics (both frames share them)
    Ri = R_of_quat(state_i[0:4]) @ Ris
    pi = state_i[4:7] + R_of_quat(state_i[0:4]) @ pis
    Rj = R_of_quat(state_j[0:4]) @ Ris
    pj = state_j[4:7] + R_of_quat(state_j[0:4]) @ pis
    vi, bgi, bai = state_i[7:10], state_i[10:13], state_i[13:16]
    vj, bgj, baj = state_j[7:10], state_j[10:13], state_j[13:16]
    dt, dR, dp, dv = rec[0], R_of_quat(rec[1:5]), rec[5:8], rec[8:11]
    sic = rec[236:461].reshape(15, 15)
    dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba = [rec[461 + 9 * k:470 + 9 * k].reshape(3, 3) for k in range(5)]
    dbg, dba = bgi - bias_lin[0:3], bai - bias_lin[3:6]
    r = np.zeros(15)
    r[0:3] = log_so3((dR @ exp_so3(dq_dbg @ dbg)).T @ Ri.T @ Rj)
    r[3:6] = Ri.T @ (pj - pi - dt * vi - 0.5 * dt * dt * GRAVITY) - (dp + dp_dbg @ dbg + dp_dba @ dba)
    r[6:9] = Ri.T @ (vj - vi - dt * GRAVITY) - (dv + dv_dbg @ dbg + dv_dba @ dba)
    r[9:12] = bgj - bgi
    r[12:15] = baj - bai
    return sic @ r


# ------------------------------------------------------------------ A12
def marginalization_residual(states, lin, S, f):
    e = np.zeros(15 * len(states))
    for i, (s, l) in enumerate(zip(states, lin)):
        e[15 * i:15 * i + 3] = log_so3(R_of_quat(l[0:4]).T @ R_of_quat(s[0:4]))
        e[15 * i + 3:15 * i + 6] = s[4:7] - l[4:7]
        e[15 * i + 6:15 * i + 15] = s[7:16] - l[7:16]
    return S @ e + f


# ------------------------------------------------------------------ vectors
def random_state(rng, scale=1.0):
    q = quat_of_R(exp_so3(scale * rng.uniform(-0.6, 0.6, 3)))
    return np.concatenate([q, rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 3), 1e-2 * rng.standard_normal(3), 5e-2 * rng.standard_normal(3)])


def main():
    rng = np.random.default_rng(20261005)
    extr = np.array([-7.7071797555374275e-03, 1.0499323370587278e-02, 7.0175280029197162e-01, 7.1230146066895372e-01, -0.0216401454975, -0.064676986768,
                     0.00981073058949, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])
    extr_imu = extr.copy()           # a second set with a non-trivial IMU extrinsic (preintegration_factor.h:43-46)
    extr_imu[7:11] = quat_of_R(exp_so3(np.array([0.02, -0.03, 0.015])))
    extr_imu[11:14] = [0.01, -0.02, 0.005]
    noise = np.concatenate([np.eye(3).ravel() * 2.8791302399999997e-08, np.eye(3).ravel() * 4.0e-6, np.eye(3).ravel() * 3.7608844899999997e-10,
                            np.eye(3).ravel() * 9.0e-6])
    W = np.array([[458.654 / np.sqrt(0.5), 0.0], [0.0, 457.296 / np.sqrt(0.5)]])
    out = dict(extr=extr, extr_imu=extr_imu, noise=noise, W=W)
    # A7: segments of 2, 3, 11 and 40 samples (a single step leaves the 9 x 9 covariance block at rank 6)
    segs, recs, pars = [], [], []
    for n in (2, 3, 11, 40):
        t0 = rng.uniform(1.0, 2.0)
        t = t0 + 0.005 * np.arange(n)
        imu = np.column_stack([t, 0.3 * rng.standard_normal((n, 3)), np.array([0.2, -0.1, 9.7]) + 0.5 * rng.standard_normal((n, 3))])
        bg, ba = 1e-2 * rng.standard_normal(3), 5e-2 * rng.standard_normal(3)
        t_end = t[-1] + 0.004
        segs.append(imu)
        pars.append(np.concatenate([[t_end], bg, ba]))
        recs.append(preintegrate(imu, t_end, bg, ba, noise))
    out["pre_imu"] = np.concatenate(segs)
    out["pre_off"] = np.cumsum([0] + [len(s) for s in segs]).astype(np.int32)
    out["pre_par"] = np.array(pars)
    out["pre_rec"] = np.array(recs)
    # A8: 64 factors between 6 frames
    nfr, nl, nf = 6, 40, 64
    states = np.array([random_state(rng, 0.3) for _ in range(nfr)])
    states[:, 4:7] *= 0.3
    z_ref = rng.standard_normal((nl, 3)) * [0.4, 0.3, 0.0] + [0, 0, 1.0]
    z_ref /= np.linalg.norm(z_ref, axis=1, keepdims=True)
    inv_depth = rng.uniform(0.1, 0.5, nl)
    tgt, ref, lm = rng.integers(0, nfr, nf), rng.integers(0, nfr, nf), rng.integers(0, nl, nf)
    z_obs = rng.standard_normal((nf, 3)) * [0.3, 0.3, 0.0] + [0, 0, 1.0]
    z_obs /= np.linalg.norm(z_obs, axis=1, keepdims=True)
    tangent = np.array([tangent_frame(z) for z in z_obs])
    out.update(rp_states=states, rp_zref=z_ref, rp_invd=inv_depth, rp_tgt=tgt.astype(np.int32), rp_ref=ref.astype(np.int32), rp_lm=lm.astype(np.int32),
               rp_zobs=z_obs, rp_tangent=tangent.reshape(nf, 9),
               rp_r=np.array([reprojection_residual(states[tgt[k]], states[ref[k]], z_ref[lm[k]], inv_depth[lm[k]], tangent[k], extr, W) for k in range(nf)]))
    # A11: the four records above between random state pairs, about perturbed linearisation biases, with both extrinsic sets
    si = np.array([random_state(rng, 0.3) for _ in range(4)])
    sj = np.array([random_state(rng, 0.3) for _ in range(4)])
    for k in range(4):   # make state j roughly consistent with the record so that the rotation residual stays well below pi
        Ri = R_of_quat(si[k, 0:4])
        dR = R_of_quat(recs[k][1:5])
        sj[k, 0:4] = quat_of_R(Ri @ dR @ exp_so3(0.05 * rng.standard_normal(3)))
    lin_b = np.column_stack([si[:, 10:13] + 1e-3 * rng.standard_normal((4, 3)), si[:, 13:16] + 1e-2 * rng.standard_normal((4, 3))])
    out.update(pf_si=si, pf_sj=sj, pf_lin=lin_b,
               pf_r=np.array([preintegration_residual(si[k], sj[k], recs[k], lin_b[k], extr) for k in range(4)]),
               pf_r_imu=np.array([preintegration_residual(si[k], sj[k], recs[k], lin_b[k], extr_imu) for k in range(4)]))
    # A12: a prior over 5 frames with a dense S
    npf = 5
    lin = np.array([random_state(rng, 0.3) for _ in range(npf)])
    cur = lin.copy()
    for i in range(npf):
        cur[i, 0:4] = quat_of_R(R_of_quat(lin[i, 0:4]) @ exp_so3(0.05 * rng.standard_normal(3)))
        cur[i, 4:16] += 0.02 * rng.standard_normal(12)
    D = 15 * npf
    S = np.triu(rng.standard_normal((D, D))) * 5.0
    f = rng.standard_normal(D)
    out.update(mp_states=cur, mp_lin=lin, mp_S=S, mp_f=f, mp_r=marginalization_residual(cur, lin, S, f))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "witness_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
