"""Pins for the oracle's restatement of the Ceres dogleg solve (PARITY UNPINNED vs Ceres itself)."""
import numpy as np

from rd_vio_amd import synth
from test_oracle_estimation import _perturb_state


def _pre(oracle):
    return lambda imu, t, bg, ba: oracle.preintegrate(imu, t, bg, ba, synth.EUROC_NOISE)


def _robust_cost(oracle, pb, states, invd):
    r, *_ = oracle.reprojection_eval(pb["tgt"], pb["ref"], pb["lm"], pb["tangent"], pb["z_ref"], invd, states,
                                     pb["extr"], pb["sqrt_inv_cov"], jac=False)
    return 0.5 * np.log1p((r * r).sum(axis=1)).sum()


def test_vision_only_converges_to_stationary_point(oracle):
    pb = synth.make_window_problem(6, 80, 4, preintegrate=_pre(oracle), with_preint=False, with_prior=False)
    pb["frame_fixed"][:2] = 1  # gauge: two fixed frames
    st, invd, sm = oracle.ba_solve(pb, 50)
    assert sm.termination == 0 and sm.final_cost < 0.5 * sm.initial_cost
    assert abs(sm.final_cost - _robust_cost(oracle, pb, st, invd)) < 1e-9 * sm.final_cost
    # finite-difference gradient of the robust cost at the solution is ~0 compared with the start
    def grad(states, d):
        g = []
        h = 1e-6
        for i in range(2, 6):
            for c in range(6):
                e = np.zeros(15)
                e[c] = h
                sp, sm_ = states.copy(), states.copy()
                sp[i] = _perturb_state(sp[i], e)
                sm_[i] = _perturb_state(sm_[i], -e)
                g.append((_robust_cost(oracle, pb, sp, d) - _robust_cost(oracle, pb, sm_, d)) / (2 * h))
        return np.array(g)
    g0, g1 = grad(pb["states"], pb["inv_depth"]), grad(st, invd)
    assert np.abs(g1).max() < 2e-2 * np.abs(g0).max()  # function-tolerance stop, not gradient-tolerance
    # fixed frames are untouched
    assert (st[:2] == pb["states"][:2]).all()


def test_noise_free_problem_recovers_truth(oracle):
    pb = synth.make_window_problem(6, 80, 4, preintegrate=_pre(oracle), with_preint=False, with_prior=False,
                                   pix_noise=0.0)
    pb["frame_fixed"][:2] = 1
    pb["states"][:2] = pb["states_true"][:2]
    st, invd, sm = oracle.ba_solve(pb, 50)
    assert sm.final_cost < 1e-8
    assert np.abs(st[:, 4:7] - pb["states_true"][:, 4:7]).max() < 1e-5
    assert np.abs(invd / pb["inv_depth_true"] - 1).max() < 1e-5


def test_window_problem_decreases_cost_and_respects_iteration_limit(oracle):
    pb = synth.make_window_problem(9, 150, 648, preintegrate=_pre(oracle))
    for lim in (0, 1, 5, 30):
        st, invd, sm = oracle.ba_solve(pb, lim)
        assert sm.iterations <= lim
        assert sm.final_cost <= sm.initial_cost
        if lim == 0:
            assert (st == pb["states"]).all() and (invd == pb["inv_depth"]).all()
    # the 1e15 prior pins the pose of frame 0 (marginalization_factor.h:27-31)
    assert np.abs(st[0, :7] - pb["states"][0, :7]).max() < 1e-9


def test_fixed_landmarks_and_frames_are_constant(oracle):
    pb = synth.make_window_problem(5, 40, 5, preintegrate=_pre(oracle), with_prior=False)
    pb["frame_fixed"][:] = [1, 1, 0, 0, 0]
    pb["lm_fixed"][::2] = 1
    st, invd, sm = oracle.ba_solve(pb, 20)
    assert (st[:2] == pb["states"][:2]).all()
    assert (invd[::2] == pb["inv_depth"][::2]).all()
    assert sm.final_cost < sm.initial_cost


def test_pose_fixed_frame_keeps_pose_and_frees_motion(oracle):
    # FT_FIX_POSE without FT_FIX_MOTION (solver.cpp:92-97; the initializer's first keyframe, initializer.cpp:82)
    pb = synth.make_window_problem(6, 60, 7, preintegrate=_pre(oracle), with_prior=False)
    pb["frame_fixed"][0] = 2
    st, invd, sm = oracle.ba_solve(pb, 20)
    assert (st[0, :7] == pb["states"][0, :7]).all()
    assert np.abs(st[0, 7:] - pb["states"][0, 7:]).max() > 0     # v / biases of that frame did move
    assert sm.final_cost < sm.initial_cost
