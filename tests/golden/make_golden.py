#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.npz.

WHAT THESE VECTORS ARE: inputs and outputs of THIS BUILD'S CPU oracle (oracle/*.c) on small seeded cases -- regression
pins for the oracle and the HIP path.  They are NOT outputs of the reference: the reference has no tests or fixtures
and cannot be built in this image ("parity unpinned"); nothing here is derived from reference
source or data files.  Inputs come from rd_vio_amd/synth.py (seeded numpy)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def generate():
    import oracle
    from rd_vio_amd import synth

    oracle.build()
    out = {}
    # ---- image side: 160 x 120 room views 50 ms apart
    W, H = 160, 120
    K = synth.EUROC_K.copy()
    K[:2] *= 160.0 / 752.0
    a = synth.render_room(*synth.traj_pose(1.0), K, W, H)
    b = synth.render_room(*synth.traj_pose(1.05), K, W, H)
    out["img_a"], out["img_b"] = a, b
    L, ia, da = oracle.preprocess(a)
    _, ib, db = oracle.preprocess(b)
    out["pyr_img_sha"] = np.array([sha(ia), sha(ib)])
    out["pyr_deriv_sha"] = np.array([sha(da), sha(db)])
    lvl0 = np.ascontiguousarray(oracle.level_view(L, ia, 0))
    kps = oracle.detect_keypoints(lvl0, np.zeros((0, 2)), 40, 8.0)
    out["detect_kps"] = kps
    nxt, st = oracle.track_keypoints(L, (ia, da), (ib, db), kps)
    out["track_next"], out["track_status"] = nxt, st
    # ---- estimation side
    rng = np.random.default_rng(7)
    seg = synth.make_imu_segment(1.0, 1.25, rng=rng, bg=synth.TRUE_BG, ba=synth.TRUE_BA)
    out["imu_seg"] = seg
    out["preint"] = oracle.preintegrate(seg, 1.25, synth.TRUE_BG * 0.5, synth.TRUE_BA * 0.5, synth.EUROC_NOISE)
    pb = synth.make_ba_problem(n_frames=4, n_landmarks=12, seed=11)
    r, Jt, Jr, Jd = oracle.reprojection_eval(pb["tgt"], pb["ref"], pb["lm"], pb["tangent"], pb["z_ref"], pb["inv_depth"], pb["states"],
                                             pb["extr"], pb["sqrt_inv_cov"])
    out["rpe_r"], out["rpe_Jt"], out["rpe_Jr"], out["rpe_Jd"] = r, Jt, Jr, Jd
    pre = lambda imu, t, bg, ba: oracle.preintegrate(imu, t, bg, ba, synth.EUROC_NOISE)  # noqa: E731
    wp = synth.make_window_problem(5, 40, 21, preintegrate=pre)
    s, d, sm = oracle.ba_solve(wp, 15)
    out["solve_states"], out["solve_invd"] = s, d
    out["solve_summary"] = np.array([sm.iterations, sm.successful_steps, sm.initial_cost, sm.final_cost, sm.termination], dtype=np.float64)
    S, f, lin, Lam, eta = oracle.marginalize(*synth.make_marg_inputs(wp))
    out["marg_Lambda"], out["marg_eta"], out["marg_StS"], out["marg_Stf"] = Lam, eta, S.T @ S, S.T @ f
    return out


if __name__ == "__main__":
    vec = generate()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    np.savez_compressed(path, **vec)
    print(path, os.path.getsize(path), "bytes")
