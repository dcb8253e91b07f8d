"""The oracle's residuals (rows A7, A8, A11, A12) against the independent numpy witness (tests/golden/make_witness.py: a second
transcription of the reference's formulas by another route -- rotation matrices, no shared helpers; vectors committed in
tests/golden/witness_vectors.npz).  Removes the single-transcriber risk of the oracle; the reference itself cannot run here, so
parity stays "unpinned" in the task's sense (oracle/ headers)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vec():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "witness_vectors.npz")))


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_witness_vectors_are_what_the_script_produces(vec, tmp_path, monkeypatch):
    """the committed fixture is reproducible from the committed generator"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_witness", os.path.join(ROOT, "tests", "golden", "make_witness.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    saved = {}
    monkeypatch.setattr(np, "savez_compressed", lambda path, **kw: saved.update(kw))
    mod.main()
    assert set(saved) == set(vec)
    for k in vec:
        assert np.allclose(vec[k], saved[k], rtol=1e-13, atol=0), k


def test_preintegrator_matches_the_witness(oracle, vec):
    off = vec["pre_off"]
    for k in range(len(off) - 1):
        imu, par = vec["pre_imu"][off[k]:off[k + 1]], vec["pre_par"][k]
        got = oracle.preintegrate(imu, par[0], par[1:4], par[4:7], vec["noise"])
        ref = vec["pre_rec"][k]
        # t, q, p, v; covariance; sqrt information; bias Jacobians -- each against its own scale
        q_sign = np.sign(got[1:5] @ ref[1:5])
        assert abs(got[0] - ref[0]) < 1e-14 and np.abs(q_sign * got[1:5] - ref[1:5]).max() < 1e-12
        for a, b in ((5, 11), (11, 236), (236, 461), (461, 506)):
            assert _rel(got[a:b], ref[a:b]) < 1e-9, (k, a, _rel(got[a:b], ref[a:b]))


def test_reprojection_residual_matches_the_witness(oracle, vec):
    r, *_ = oracle.reprojection_eval(vec["rp_tgt"], vec["rp_ref"], vec["rp_lm"], vec["rp_tangent"], vec["rp_zref"], vec["rp_invd"], vec["rp_states"],
                                     vec["extr"], vec["W"])
    assert _rel(r, vec["rp_r"]) < 1e-12
    for k in range(len(vec["rp_zobs"])):   # and the tangent frames
        assert np.abs(np.asarray(oracle.tangent_frame(vec["rp_zobs"][k])).ravel() - vec["rp_tangent"][k]).max() < 1e-14


def test_preintegration_factor_residual_matches_the_witness(oracle, vec):
    for key, extr in (("pf_r", vec["extr"]), ("pf_r_imu", vec["extr_imu"])):
        for k in range(4):
            r, _, _ = oracle.preintegration_eval(vec["pf_si"][k], vec["pf_sj"][k], vec["pre_rec"][k], vec["pf_lin"][k], extr)
            assert _rel(r, vec[key][k]) < 1e-9, (key, k, _rel(r, vec[key][k]))


def test_marginalization_residual_matches_the_witness(oracle, vec):
    r, _ = oracle.marginalization_eval(vec["mp_states"], vec["mp_lin"], vec["mp_S"], vec["mp_f"])
    assert _rel(r, vec["mp_r"]) < 1e-12


# ---- the HIP entries against the same witness vectors
@pytest.mark.gpu
def test_hip_entries_match_the_witness(vec):
    import rd_vio_amd

    with rd_vio_amd.Context(max_width=752, max_height=480, max_features=256, max_window=8, max_factors=2048) as ctx:
        off = vec["pre_off"]
        segs = [vec["pre_imu"][off[k]:off[k + 1]] for k in range(len(off) - 1)]
        got = ctx.preintegrate(segs, vec["pre_par"][:, 0], vec["pre_par"][:, 1:4], vec["pre_par"][:, 4:7], vec["noise"])
        for k, ref in enumerate(vec["pre_rec"]):
            q_sign = np.sign(got[k][1:5] @ ref[1:5])
            assert abs(got[k][0] - ref[0]) < 1e-14 and np.abs(q_sign * got[k][1:5] - ref[1:5]).max() < 1e-12
            for a, b in ((5, 11), (11, 236), (236, 461), (461, 506)):
                assert _rel(got[k][a:b], ref[a:b]) < 1e-9, (k, a)
        pb = dict(states=vec["rp_states"], extr=vec["extr"], sqrt_inv_cov=vec["W"], z_ref=vec["rp_zref"], inv_depth=vec["rp_invd"],
                  tgt=vec["rp_tgt"], ref=vec["rp_ref"], lm=vec["rp_lm"], tangent=vec["rp_tangent"])
        order = np.argsort(vec["rp_lm"], kind="stable")
        r, *_ = ctx.reprojection_eval(pb)
        assert _rel(r, vec["rp_r"]) < 1e-12
        del order
        # A11 through the solver's own evaluation (rdvio_hip_ba_linearize): two free frames, one preintegration factor
        for key, extr in (("pf_r", vec["extr"]), ("pf_r_imu", vec["extr_imu"])):
            for k in range(4):
                st = np.stack([vec["pf_si"][k], vec["pf_sj"][k]])
                lin_states = st.copy()
                lin_states[0, 10:16] = vec["pf_lin"][k]
                pbk = dict(states=st, extr=extr, sqrt_inv_cov=vec["W"], z_ref=np.zeros((0, 3)), inv_depth=np.zeros(0), tgt=[], ref=[], lm=[],
                           tangent=np.zeros((0, 9)), pre_i=[0], pre_j=[1], preint=vec["pre_rec"][k:k + 1], frame_fixed=np.zeros(2, dtype=np.uint8))
                lin = ctx.ba_linearize(pbk, lin_states, robust_loss=False)
                assert _rel(lin["r_preint"][0], vec[key][k]) < 1e-9, (key, k)
        # A12: five free frames under the prior alone
        n = len(vec["mp_states"])
        pbm = dict(states=vec["mp_states"], extr=vec["extr"], sqrt_inv_cov=vec["W"], z_ref=np.zeros((0, 3)), inv_depth=np.zeros(0), tgt=[], ref=[], lm=[],
                   tangent=np.zeros((0, 9)), prior_frames=np.arange(n, dtype=np.int32), lin=vec["mp_lin"], S=vec["mp_S"], f=vec["mp_f"],
                   frame_fixed=np.zeros(n, dtype=np.uint8))
        lin = ctx.ba_linearize(pbm, None, robust_loss=False)
        assert _rel(lin["r_prior"], vec["mp_r"]) < 1e-12
