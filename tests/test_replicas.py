"""N>1 path of bench.py on CPU: two gloo ranks, replicas only (no data-path collective), max-over-ranks timing."""
import os
import time

import pytest
import torch
import torch.multiprocessing as mp

from rd_vio_amd import replica


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w, dist = replica.init_distributed("gloo")
    count = [0]

    def step(k):
        count[0] += 1
        time.sleep(0.002 * (1 + 2 * rank))  # rank 1 is 3x slower

    el = replica.timed_region(step, 20, sync=lambda: None, dist=dist, first_index=5)
    q.put((rank, w, count[0], el, replica.aggregate_rate(w, 20, el)))
    dist.destroy_process_group()


def test_two_replicas_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, c0, e0, v0), (r1, w1, c1, e1, v1) = res
    assert (w0, w1) == (2, 2) and c0 == 20 and c1 == 20       # exactly K steps on each replica
    assert abs(e0 - e1) < 1e-9                                   # both report the MAX over ranks
    assert e0 >= 20 * 0.006 * 0.9                                # ... which is the slow rank's time
    assert abs(v0 - 2 * 20 / e0) < 1e-9                          # whole-job aggregate: N*K / max time


def test_single_process_path():
    os.environ.pop("WORLD_SIZE", None)
    r, lr, w, dist = replica.init_distributed()
    assert (r, w, dist) == (0, 1, None)
    n = [0]
    el = replica.timed_region(lambda k: n.__setitem__(0, n[0] + 1), 7, sync=lambda: None)
    assert n[0] == 7 and el >= 0
