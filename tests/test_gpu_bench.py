"""bench.py's multi-sequence leg: independent sequences sharing one GPU (one context and one host thread each, with three lanes
or one, stepped from Python or through the native rdvio_hip_frame_step) must not see each other -- every sequence's solver / marginalisation results are bit-identical to those of a sequence that
ran alone.  (The library keeps its state per context: SURVEY F9's process-global state is not reproduced.)"""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _results(seq, n_kp=None):
    s0, d0, sm0 = seq.ctx.ba_fetch(0)
    s1, d1, sm1 = seq.ctx.ba_fetch(1)
    n = seq.n_out.value if n_kp is None else n_kp   # (the native frame step keeps its keypoint count to itself)
    return (s0.copy(), d0.copy(), sm0.iterations, sm0.successful_steps, sm0.final_cost, s1.copy(), sm1.iterations, sm1.final_cost,
            seq.kp_buf[:n].copy())


def test_concurrent_sequences_do_not_interact():
    import torch

    import bench

    cfg = dict(bench.CONFIGS["euroc_v101"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    alone = bench.Sequence(cfg, torch, dev, 0, overlap=True, max_factors=4096)
    for k in range(6):
        alone.step(k)
    alone.ctx.sync()
    ref = _results(alone)
    host = alone.wl
    seqs = [bench.Sequence(cfg, torch, dev, 0, overlap=(i % 2 == 0), host=host, max_factors=4096) for i in range(4)]
    errs = []

    def run(sq, native):
        try:
            if native:   # rdvio_hip_frame_step: the whole frame as one C call (what rdvio_hip_run_sequences drives)
                d = sq.frame_step_desc()
                for k in range(6):
                    sq.ctx._check(sq.ctx._lib.rdvio_hip_frame_step(ctypes.byref(d), k))
            else:
                for k in range(6):
                    sq.step(k)
            sq.ctx.sync()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=run, args=(sq, i >= 2)) for i, sq in enumerate(seqs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert ref[2] > 0 and len(ref[8]) > 0
    for sq in seqs:
        got = _results(sq, len(ref[8]))
        for a, b in zip(ref, got):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        sq.ctx.close()
    alone.ctx.close()


def test_multi_sequence_leg_reports_an_aggregate():
    import torch

    import bench

    cfg = dict(bench.CONFIGS["euroc_v101"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    first = bench.Sequence(cfg, torch, dev, 0, overlap=True, max_factors=4096)
    rep = bench.multi_sequence(cfg, torch, dev, 0, first.wl, n_seq=3, steps=8, warmup=2)
    first.ctx.close()
    assert "error" not in rep, rep
    assert rep["sequences"] == 3 and rep["aggregate_fps"] > 0 and abs(rep["aggregate_fps"] - 3 * rep["per_sequence_fps"]) <= 1.0
    assert rep["window_solve_iterations"] == [30]   # every sequence's last window solve ran the bench problem to the limit
