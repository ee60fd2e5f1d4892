"""world_size-2 gloo test of the N>1 path bench.py uses (frame sharding + barrier + MAX over ranks). CPU only."""
import json
import os
import socket

import torch
import torch.multiprocessing as mp

from bevfusion_multimodal_3d_object_detection_amd import replicas


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist = replicas.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    frames = replicas.shard_frames(11, rank, world)
    replicas.barrier(dist)
    elapsed = replicas.max_over_ranks(1.0 + rank, dist, torch.device("cpu"))       # rank 1 is the slow one
    # every rank regenerates its own frames from (seed, index): check two ranks never share a seed
    seeds = [replicas.frame_seed(0x5EED, 2, f) for f in frames]
    q.put((rank, frames, elapsed, seeds))
    replicas.barrier(dist)
    dist.destroy_process_group()


def test_two_rank_replicas_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_frames = sorted(f for _, fr, _, _ in got for f in fr)
    assert all_frames == list(range(11))                                  # a partition, nothing lost or duplicated
    assert all(e == 2.0 for _, _, e, _ in got)                            # MAX over ranks, identical everywhere
    assert len({s for _, _, _, ss in got for s in ss}) == 11
    assert replicas.aggregate_fps(40, 2, 2.0) == 40.0


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist = replicas.init("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
    x = torch.randn(8, 16, generator=torch.Generator().manual_seed(100 + rank))     # each rank its own shard
    net(x).square().mean().backward()
    local = [p.grad.clone() for p in net.parameters()]
    n = replicas.allreduce_gradients(net.parameters(), dist, bucket_bytes=1024)     # small buckets: several collectives
    q.put((rank, [g.numpy() for g in local], [p.grad.numpy().copy() for p in net.parameters()], n))
    replicas.barrier(dist)
    dist.destroy_process_group()


def test_dp_gradient_allreduce_is_mean_of_rank_gradients():
    """N-rank all-reduced grads == mean of the N single-rank grads (loss normalisers stay per shard)."""
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, r0, n0), (_, l1, r1, n1) = got
    assert n0 == n1 and n0 >= 2
    for a, b, ra, rb in zip(l0, l1, r0, r1):
        assert np.allclose(ra, (a + b) / 2, rtol=1e-6, atol=1e-7) and np.array_equal(ra, rb)


def _reducer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from bevfusion_multimodal_3d_object_detection_amd import training
    dist = replicas.init("gloo")
    g = torch.Generator().manual_seed(7 + rank)
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in [(5, 3), (7,), (2, 2, 2), (11,)]]
    local = [torch.randn(*p.shape, generator=g) for p in params]
    red = replicas.GradReducer(dist)
    sink = training.GradSink(red)
    sink.add(params[0], local[0])
    sink.add(params[1], local[1] * 0.5)
    sink.add(params[1], local[1] * 0.5)              # repeated contribution, summed before it travels
    sink.ready()                                     # first bucket leaves while "backward" goes on
    sink.add(params[2], local[2])
    sink.ready()
    sink.add(params[3], local[3])
    sink.finish()                                    # third bucket + wait + average
    q.put((rank, [t.numpy() for t in local], [sink.get(p).detach().numpy().copy() for p in params], red.collectives))
    replicas.barrier(dist)
    dist.destroy_process_group()


def test_overlapped_reducer_averages_each_bucket():
    """The in-backward reducer (async all-reduce per group of final gradients) gives the mean over ranks."""
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, r0, n0), (_, l1, r1, n1) = got
    assert n0 == n1 == 3
    for a, b, ra, rb in zip(l0, l1, r0, r1):
        assert ra.shape == a.shape and np.allclose(ra, (a + b) / 2, rtol=1e-6, atol=1e-7) and np.array_equal(ra, rb)


def test_single_process_is_a_noop():
    assert replicas.allreduce_gradients([], None) == 0
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert replicas.init("gloo") is None
    assert replicas.max_over_ranks(3.5, None, torch.device("cpu")) == 3.5
    assert replicas.shard_frames(5, 0, 1) == [0, 1, 2, 3, 4]


def _run_bench(argv, extra_env=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BEVF_BENCH_STUB="1", BEVF_DIST_BACKEND="gloo", **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd="/tmp")
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (no torchrun) must run 2 ranks -- the parent spawns them before any
    GPU call -- and rank 0 prints ONE line with n_gpus 2.  CPU rehearsal: gloo + a stub step (BEVF_BENCH_STUB=1)."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3
    assert lines[0]["dist"] == {"backend": "gloo", "ranks": 2}


def test_bench_single_rank_and_world_mismatch():
    r, lines = _run_bench(["--steps", "2"])
    assert r.returncode == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1
    # a launcher environment whose size differs from --gpus is refused (never a silent n_gpus that differs from the flag)
    r, lines = _run_bench(["--gpus", "4", "--steps", "2"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and not lines and "--gpus 4" in r.stderr


def test_exit_guard_ends_a_stuck_rank_with_a_failure_status():
    """bench.py at N > 1: a hung extra leg (a collective that never completes) ends the rank with a NON-ZERO status and an error
    record on stderr -- a hang must reach the launcher as a failure; a leg that completes cancels the guard."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench; bench.install_exit_guard(0.5); "
            "time.sleep(60); sys.exit(0)" % root)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=50)
    assert r.returncode != 0 and r.returncode == 3 and time.time() - t0 < 40
    rec = [l for l in r.stderr.splitlines() if l.startswith("[bench extra] ")]
    assert rec and "deadline" in json.loads(rec[0][len("[bench extra] "):])["error"]
    code = ("import sys, time; sys.path.insert(0, %r); import bench; g = bench.install_exit_guard(0.5); g.cancel(); "
            "time.sleep(1.5); sys.exit(0)" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=50)
    assert r.returncode == 0 and "deadline" not in r.stderr
