"""GPU parity (MI355X): HIP image path vs the CPU oracle, bit-exact (integer / index work).
All calls go through the C ABI (rd_vio_amd.binding -> librdvio_hip.so)."""
import numpy as np
import pytest

import rd_vio_amd
from rd_vio_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = rd_vio_amd.Context(max_width=1280, max_height=720, max_features=2048)
    yield c
    c.close()


def _arena_equal(oracle, L, a, b, kind):
    """compare only the bytes the layout defines (row padding beyond w+2B is unspecified)"""
    for lv in range(L.levels):
        va = (oracle.level_view if kind == "img" else oracle.deriv_view)(L, a, lv, with_border=True)
        vb = (oracle.level_view if kind == "img" else oracle.deriv_view)(L, b, lv, with_border=True)
        if not (va == vb).all():
            bad = np.argwhere(va != vb)
            return False, (lv, bad[:5].tolist(), int((va != vb).sum()))
    return True, None


@pytest.mark.parametrize("size,seed", [((752, 480), 1), ((1280, 720), 2), ((750, 477), 3), ((100, 90), 4)])
def test_preprocess_bit_exact(ctx, oracle, size, seed):
    w, h = size
    img = synth.render_scene(w, h, seed=seed)
    Lo, pi, pd = oracle.preprocess(img)
    g = rd_vio_amd.HipImage(ctx, 0, img)
    g.preprocess(6.0, 8, 8)
    gi, gd = g.download()
    ok, info = _arena_equal(oracle, Lo, pi, gi, "img")
    assert ok, ("image arena mismatch", info)
    ok, info = _arena_equal(oracle, Lo, pd, gd, "deriv")
    assert ok, ("deriv arena mismatch", info)


def test_preprocess_edge_images(ctx, oracle):
    for img in (np.zeros((480, 752), np.uint8), np.full((480, 752), 255, np.uint8),
                np.random.default_rng(0).integers(0, 256, (480, 752), dtype=np.uint8)):
        Lo, pi, pd = oracle.preprocess(img)
        g = rd_vio_amd.HipImage(ctx, 1, img)
        g.preprocess(6.0, 8, 8)
        gi, gd = g.download()
        assert _arena_equal(oracle, Lo, pi, gi, "img")[0]
        assert _arena_equal(oracle, Lo, pd, gd, "deriv")[0]


def _pair(ctx, oracle, w, h, seed, offset, rot=0.0):
    a = synth.render_scene(w, h, seed=seed)
    b = synth.render_scene(w, h, seed=seed, offset=offset, rot=rot)
    La, ia, da = oracle.preprocess(a)
    Lb, ib, db = oracle.preprocess(b)
    ga, gb = rd_vio_amd.HipImage(ctx, 0, a), rd_vio_amd.HipImage(ctx, 1, b)
    ga.preprocess()
    gb.preprocess()
    return La, (ia, da), (ib, db), ga, gb


@pytest.mark.parametrize("w,h,n,offset,rot", [
    (752, 480, (15, 10), (3.3, -2.1), 0.0),
    (752, 480, (20, 15), (-7.6, 5.2), 0.01),
    (1280, 720, (40, 25), (2.2, 1.7), 0.004),
])
def test_track_keypoints_bit_exact(ctx, oracle, w, h, n, offset, rot):
    L, A, B, ga, gb = _pair(ctx, oracle, w, h, 11, offset, rot)
    pts = synth.jittered_grid(w, h, n[0], n[1])
    rng = np.random.default_rng(5)
    guess = pts - np.array(offset) + rng.uniform(-3, 3, pts.shape)
    for g in (None, guess):
        ref_next, ref_st = oracle.track_keypoints(L, A, B, pts, guess=g)
        got_next, got_st = ga.track_keypoints(gb, pts, g)
        assert (ref_st == got_st).all(), np.argwhere(ref_st != got_st)[:10]
        assert ref_st.sum() > 0.7 * len(pts)
        m = ref_st > 0
        assert (ref_next[m] == got_next[m]).all()  # bit-exact positions
        # non-survivors are left untouched (opencv_image.cpp:148-153)
        expect = g if g is not None else np.zeros_like(pts)
        assert (got_next[~m] == expect[~m]).all()


def test_lk_flow_bit_exact_including_failures(ctx, oracle):
    L, A, B, ga, gb = _pair(ctx, oracle, 752, 480, 12, (4.0, 3.0))
    rng = np.random.default_rng(9)
    pts = rng.uniform(-30, 790, (300, 2)).astype(np.float32)   # includes points outside the image
    pts[:, 1] = rng.uniform(-30, 510, 300)
    init = pts + rng.uniform(-5, 5, pts.shape).astype(np.float32)
    ref_next, ref_st = oracle.lk_flow(L, A[0], A[1], B[0], pts, init)
    got_next, got_st = ga.lk_flow(gb, pts, init)
    assert (ref_st == got_st).all()
    assert (ref_next == got_next).all()
    assert 0 < ref_st.sum() < len(pts)


def test_track_empty_and_flat(ctx, oracle):
    flat = np.full((480, 752), 128, np.uint8)
    g0, g1 = rd_vio_amd.HipImage(ctx, 0, flat), rd_vio_amd.HipImage(ctx, 1, flat)
    g0.preprocess()
    g1.preprocess()
    nxt, st = g0.track_keypoints(g1, np.zeros((0, 2)))
    assert len(st) == 0
    nxt, st = g0.track_keypoints(g1, np.array([[100.0, 100.0], [300.0, 200.0]]))
    assert (st == 0).all()


@pytest.mark.parametrize("size,seed", [((752, 480), 21), ((1280, 720), 22)])
def test_harris_and_detect_bit_exact(ctx, oracle, size, seed):
    w, h = size
    img = synth.render_scene(w, h, seed=seed)
    Lo, pi, pd = oracle.preprocess(img)
    lvl0 = np.ascontiguousarray(oracle.level_view(Lo, pi, 0))
    g = rd_vio_amd.HipImage(ctx, 0, img)
    g.preprocess()
    ref = oracle.harris_response(lvl0)
    got = g.harris_response()
    assert (ref.view(np.uint32) == got.view(np.uint32)).all()
    for existing in (np.zeros((0, 2)), oracle.detect_keypoints(lvl0, np.zeros((0, 2)), 40, 20.0)):
        for maxp, dist in ((150, 10.0), (300, 20.0), (1000, 10.0)):
            ref_kp = oracle.detect_keypoints(lvl0, existing, maxp, dist)
            got_kp = g.detect_keypoints(existing, maxp, dist)
            assert ref_kp.shape == got_kp.shape and (ref_kp == got_kp).all()
            assert len(ref_kp) > len(existing)


@pytest.mark.parametrize("kind", ["room_752", "room_1280", "fractional_existing", "radius_above_min_distance", "noise_host_road", "flat",
                                  "top_bins_suffice", "top_bins_fall_short"])
def test_device_keypoint_selection_is_order_exact(ctx, oracle, kind, monkeypatch):
    """Device-side GFTT ordering + greedy minDistance + Poisson-disk thinning + border test (select_kernels.hip) against the
    oracle AND against the product's own host road (RDVIO_HOST_SELECT=1, host_select.cpp): same keypoints, same order, bit
    for bit.  Cases: ray-cast room frames, existing keypoints at fractional (tracked) positions incl. two in one Poisson cell,
    a Poisson radius above GFTT's minDistance (the inserts then depend on each other), a noise image with > 8192 local maxima
    (beyond the kernels' LDS capacity: the entry point must take the host road and still be exact), a flat image; two textures
    with thousands of maxima for the sort that starts with the top response bins -- one where their maxima yield maxCorners
    corners (~6800 maxima, 1026 sorted) and one where the strongest maxima crowd into one region, the greedy pass over them ends
    short and the kernel has to start over with every candidate (both roads asserted through rdvio_hip_debug_last_select_path)."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(5)
    w, h, maxp, dist = 752, 480, 150, 10.0
    if kind == "room_1280":
        w, h, maxp = 1280, 720, 1000
        K = np.array([[900.0, 0, 640.0], [0, 900.0, 360.0], [0, 0, 1.0]])
        img = synth.make_stream(1, w, h, K)[0][0]
    elif kind == "noise_host_road":
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    elif kind == "flat":
        img = np.full((h, w), 77, np.uint8)
    elif kind == "top_bins_suffice":
        g7 = np.random.default_rng(7)
        tex = ndi.gaussian_filter(g7.normal(0, 1, (h, w)), 3.5)
        img = (128 + 90 * tex / np.abs(tex).max()).astype(np.uint8)
    elif kind == "top_bins_fall_short":
        g7 = np.random.default_rng(7)
        fine = ndi.gaussian_filter(g7.normal(0, 1, (h, w)), 2.0)
        coarse = ndi.gaussian_filter(g7.normal(0, 1, (h, w)), 3.0)
        mix = 0.45 * coarse / coarse.std()
        mix[100:380, 200:480] = (fine / fine.std())[100:380, 200:480]
        img = np.clip(128 + 30 * mix, 0, 255).astype(np.uint8)
        maxp = 250
    else:
        img = synth.make_stream(2, w, h, synth.EUROC_K)[0][1]
    Lo, pi, pd = oracle.preprocess(img)
    lvl0 = np.ascontiguousarray(oracle.level_view(Lo, pi, 0))
    g = rd_vio_amd.HipImage(ctx, 0, img)
    g.preprocess()
    existing = np.zeros((0, 2))
    if kind == "fractional_existing":
        first = oracle.detect_keypoints(lvl0, np.zeros((0, 2)), 60, 20.0)
        existing = first + rng.uniform(-3.0, 3.0, first.shape)
        existing = np.concatenate([existing, existing[:5] + 0.5, [[-4.0, 100.25], [w + 3.0, 50.0]]])   # same-cell pairs, points off the image
    if kind == "radius_above_min_distance":
        dist = 33.0
    ref_kp = oracle.detect_keypoints(lvl0, existing, maxp, dist)
    got_kp = g.detect_keypoints(existing, maxp, dist)
    assert ref_kp.shape == got_kp.shape and (ref_kp == got_kp).all()
    path = ctx._lib.rdvio_hip_debug_last_select_path(ctx._h)
    if kind == "top_bins_suffice":
        assert path == 1
    elif kind == "top_bins_fall_short":
        assert path == 2
    elif kind == "noise_host_road":
        assert path == -1
    if kind == "flat":
        assert len(got_kp) == len(existing)
    else:
        assert len(got_kp) > len(existing)
    # the product's host road gives the same answer
    host_ctx_env = dict(RDVIO_HOST_SELECT="1")
    monkeypatch.setenv("RDVIO_HOST_SELECT", "1")
    hc = rd_vio_amd.Context(max_width=w, max_height=h, max_features=64)
    try:
        gh = rd_vio_amd.HipImage(hc, 0, img)
        gh.preprocess()
        host_kp = gh.detect_keypoints(existing, maxp, dist)
    finally:
        hc.close()
    assert host_kp.shape == got_kp.shape and (host_kp == got_kp).all()


def test_capacity_and_argument_errors(ctx):
    small = rd_vio_amd.Context(max_width=128, max_height=96, max_features=8)
    img = synth.render_scene(752, 480, seed=1)
    g = rd_vio_amd.HipImage(small, 0, img)
    with pytest.raises(rd_vio_amd.RdvioError):
        g.preprocess()  # image larger than the context
    ok = rd_vio_amd.HipImage(small, 0, img[:96, :128])
    ok.preprocess()
    with pytest.raises(rd_vio_amd.RdvioError):
        ok.track_keypoints(rd_vio_amd.HipImage(small, 1, img[:96, :128]), np.zeros((4, 2)) + 50)  # slot 1 never preprocessed
    with pytest.raises(rd_vio_amd.RdvioError):
        ok.track_keypoints(ok, np.zeros((9, 2)) + 50)  # more features than capacity
    small.close()
