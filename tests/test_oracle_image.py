"""Pins for the oracle's image side (OpenCV arithmetic restated; PARITY UNPINNED vs OpenCV itself --
no OpenCV in the image -- so the pins are analytic: known shifts, hand-computed kernels, invariants)."""
import numpy as np

from rd_vio_amd import synth


def test_pyramid_levels_and_pyrdown_constant(oracle):
    img = np.full((480, 752), 77, dtype=np.uint8)
    L, pi, pd = oracle.build_pyramid(img)
    assert L.levels == 4
    assert [L.w[i] for i in range(4)] == [752, 376, 188, 94]
    assert [L.h[i] for i in range(4)] == [480, 240, 120, 60]
    for lv in range(4):
        assert (oracle.level_view(L, pi, lv, with_border=True) == 77).all()
        assert (oracle.deriv_view(L, pd, lv, with_border=True) == 0).all()


def test_pyramid_stops_at_window(oracle):
    # buildOpticalFlowPyramid stops when the next level would be <= winSize (21)
    img = np.zeros((90, 100), dtype=np.uint8)
    L, _, _ = oracle.build_pyramid(img)
    assert L.levels == 3 and L.w[2] == 25 and L.h[2] == 23


def test_scharr_on_ramp(oracle):
    # I(x,y) = 2x + 3y -> dx = (3+10+3)*2*2 = 64, dy = 16*3*2 = 96 in the interior
    ys, xs = np.mgrid[0:40, 0:60]
    img = (2 * xs + 3 * ys).clip(0, 255).astype(np.uint8)
    assert img.max() < 255
    L, pi, pd = oracle.build_pyramid(img, max_level=0)
    d = oracle.deriv_view(L, pd, 0)
    assert (d[1:-1, 1:-1, 0] == 64).all() and (d[1:-1, 1:-1, 1] == 96).all()
    # reflect-101 at the border: dx at x=0 is zero
    assert (d[:, 0, 0] == 0).all() and (d[0, :, 1] == 0).all()
    # image border is BORDER_REFLECT_101, derivative border is zero
    full = oracle.level_view(L, pi, 0, with_border=True)
    B = L.border
    assert (full[B:B + 40, B - 3] == img[:, 3]).all() and (full[B - 5, B:B + 60] == img[5, :]).all()
    assert (oracle.deriv_view(L, pd, 0, with_border=True)[:B] == 0).all()


def test_pyrdown_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (61, 75), dtype=np.uint8)
    L, pi, _ = oracle.build_pyramid(img, max_level=1)
    got = oracle.level_view(L, pi, 1)
    k = np.array([1, 4, 6, 4, 1])
    pad = np.pad(img.astype(np.int64), 2, mode="reflect")
    h, w = (61 + 1) // 2, (75 + 1) // 2
    ref = np.zeros((h, w), dtype=np.int64)
    for y in range(h):
        for x in range(w):
            win = pad[2 * y:2 * y + 5, 2 * x:2 * x + 5]
            ref[y, x] = (k[:, None] * k[None, :] * win).sum()
    assert (got == ((ref + 128) >> 8)).all()


def test_clahe_properties(oracle):
    img = synth.render_scene(752, 480, seed=3)
    out = oracle.clahe(img)
    assert out.shape == img.shape and out.std() > img.std()  # contrast is stretched
    # constant image: every tile has one full bin -> clipped + redistributed -> monotone LUT, constant output
    c = oracle.clahe(np.full((480, 752), 100, dtype=np.uint8))
    assert (c == c[0, 0]).all()
    # monotone in intensity within one image location's LUT: brighter input never maps darker (same tile mix)
    a = img.copy()
    b = np.minimum(a.astype(int) + 0, 255).astype(np.uint8)
    assert (oracle.clahe(a) == oracle.clahe(b)).all()
    # non-multiple size exercises the reflect-101 padding path
    odd = oracle.clahe(img[:477, :750])
    assert odd.shape == (477, 750)


def test_lk_recovers_known_shift(oracle):
    # the same planar texture sampled at +(3.3, -2.1): content moves by (-3.3, +2.1); no CLAHE
    a = synth.render_scene(320, 240, seed=5)
    b = synth.render_scene(320, 240, seed=5, offset=(3.3, -2.1))
    La, ia, da = oracle.build_pyramid(a)
    Lb, ib, db = oracle.build_pyramid(b)
    pts = synth.jittered_grid(320, 240, 8, 6, margin=50)
    nxt, st = oracle.track_keypoints(La, (ia, da), (ib, db), pts)
    assert st.sum() >= 0.9 * len(pts)
    flow = (nxt - pts)[st > 0]
    assert np.abs(np.median(flow, axis=0) - [-3.3, 2.1]).max() < 0.1
    # with a good initial guess it still converges to the same place
    nxt2, st2 = oracle.track_keypoints(La, (ia, da), (ib, db), pts, guess=pts + [-3.0, 2.0])
    both = (st > 0) & (st2 > 0)
    assert np.abs(nxt2[both] - nxt[both]).max() < 0.2
    # survivors only are written back; failures keep the caller's value
    assert np.allclose(nxt[st == 0], 0)


def test_lk_rejects_flat_and_border(oracle):
    flat = np.full((240, 320), 128, dtype=np.uint8)
    L, i0, d0 = oracle.build_pyramid(flat)
    nxt, st = oracle.lk_flow(L, i0, d0, i0, np.array([[100.0, 100.0]]), np.array([[100.0, 100.0]]))
    assert st[0] == 0  # minEig < 1e-4
    a = synth.render_scene(320, 240, seed=6)
    La, ia, da = oracle.build_pyramid(a)
    # points within 20 px of the border are rejected by the reference's own check
    pts = np.array([[10.0, 120.0], [160.0, 8.0], [160.0, 120.0]])
    nxt, st = oracle.track_keypoints(La, (ia, da), (ia, da), pts)
    assert list(st) == [0, 0, 1]
    assert np.abs(nxt[2] - pts[2]).max() < 1e-2
    # far outside the image at level 0 -> status 0
    nxt, st = oracle.lk_flow(La, ia, da, ia, np.array([[-100.0, 50.0]]), np.array([[-100.0, 50.0]]))
    assert st[0] == 0


def test_harris_and_gftt(oracle):
    img = np.full((120, 160), 50, dtype=np.uint8)
    img[40:80, 60:100] = 200  # a bright square: 4 corners
    resp = oracle.harris_response(img)
    assert resp[60, 20] == 0 and resp[60, 60] < 0  # flat -> 0, edge -> negative
    xy, r = oracle.good_features(img, 10, min_dist=10.0)
    assert len(xy) == 4
    corners = {(60, 40), (99, 40), (60, 79), (99, 79)}
    for (x, y) in xy:
        assert min(abs(x - cx) + abs(y - cy) for cx, cy in corners) <= 2
    assert (np.diff(r) <= 0).all()
    # minDistance suppression
    xy2, _ = oracle.good_features(img, 10, min_dist=60.0)
    assert len(xy2) < 4


def test_detect_keypoints_poisson_and_border(oracle):
    img = synth.render_scene(752, 480, seed=7)
    kp = oracle.detect_keypoints(img, np.zeros((0, 2)), 150, 20.0)
    assert 20 < len(kp) <= 150
    assert (kp[:, 0] >= 20).all() and (kp[:, 0] < 732).all() and (kp[:, 1] >= 20).all() and (kp[:, 1] < 460).all()
    d = np.linalg.norm(kp[:, None] - kp[None], axis=-1) + np.eye(len(kp)) * 1e9
    assert d.min() >= 20.0
    # existing points are kept in front and suppress new ones nearby
    kp2 = oracle.detect_keypoints(img, kp[:10], 150, 20.0)
    assert np.allclose(kp2[:10], kp[:10])
    d2 = np.linalg.norm(kp2[10:, None] - kp[None, :10], axis=-1)
    assert d2.min() >= 20.0 - 1e-9


def test_xcd_tile_order_is_a_permutation():
    """rd_vio_amd/csrc/image_kernels.hip, xcd_tile(): blocks are dealt round-robin over 8 XCDs; block b takes tile
    (b % 8) * (nb // 8) + b // 8 (the last nb % 8 blocks keep theirs).  Every tile must be taken exactly once, and the
    blocks of one XCD (equal b % 8) must take a contiguous run of tiles -- for the grids of every image size the tests use."""
    for (w, h) in ((752, 480), (1280, 720), (750, 477), (100, 90), (64, 48)):
        for border in (0, 32):
            for lv in range(4):
                lw, lh = max(1, (w + (1 << lv) - 1) >> lv), max(1, (h + (1 << lv) - 1) >> lv)
                gx, gy = (lw + 2 * border + 31) // 32, (lh + 2 * border + 7) // 8
                nb, per = gx * gy, (gx * gy) // 8
                b = np.arange(nb)
                t = np.where(b < per * 8, (b % 8) * per + b // 8, b)
                assert np.array_equal(np.sort(t), b)
                for x in range(8):
                    run = np.sort(t[(b % 8 == x) & (b < per * 8)])
                    assert len(run) == per and (per == 0 or np.array_equal(run, np.arange(run[0], run[0] + per)))
