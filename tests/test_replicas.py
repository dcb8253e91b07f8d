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


def test_bench_launcher_runs_n_replicas(tmp_path):
    """`bench.py --gpus 2` outside torchrun: the parent starts two child processes (never re-execs itself), the children
    rendezvous over gloo on 127.0.0.1, rank 0's line reports n_gpus = 2, the whole-job aggregate (2 K steps / max time) and
    one rate per replica.  --stub-step-ms replaces the GPU step by a sleep so that this runs without a GPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2", "--stub-step-ms", "3"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # ONE JSON line for the whole job
    rep = json.loads(lines[0])
    assert rep["n_gpus"] == 2 and rep["steps"] == 20 and rep["warmup"] == 2 and rep["scaling"] == "weak"
    assert len(rep["per_replica_steps_per_s"]) == 2
    # rank 1's step is twice as long: the job rate is 2 K / (rank 1's time), i.e. about rank 0's own rate
    slow = min(rep["per_replica_steps_per_s"])
    assert rep["value"] <= 2 * slow * 1.05 and rep["value"] >= 2 * slow * 0.6
    assert abs(rep["ms_per_step"] - 1e3 * 2 / rep["value"]) < 1e-2 * rep["ms_per_step"] + 1e-3


def test_bench_single_replica_stub():
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "1", "--stub-step-ms", "1"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1
