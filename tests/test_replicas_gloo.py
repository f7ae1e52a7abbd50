"""CPU, world_size 2 over gloo: the N > 1 control path of bench.py (rank discovery, volume sharding, timing barrier,
MAX / SUM reductions).  The data path itself has no collective to test -- replicas are independent."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from u2tokenizer_amd import replicas
    dist, r, w = replicas.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist is not None
    mine = replicas.shard_indices(11, r, w)
    replicas.barrier(dist)
    elapsed = 1.0 + rank  # rank 1 is the slow one
    tmax = replicas.max_over_ranks(dist, elapsed)
    total = replicas.sum_over_ranks(dist, float(len(mine)))
    q.put((rank, mine, tmax, total))
    replicas.barrier(dist)
    dist.destroy_process_group()


def test_two_replicas_shard_and_time():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, n0), (r1, s1, t1, n1) = res
    assert sorted(s0 + s1) == list(range(11)) and not set(s0) & set(s1)
    assert t0 == t1 == 2.0          # MAX over ranks: the job is as slow as its slowest replica
    assert n0 == n1 == 11.0         # whole-job unit count = sum over ranks


def test_single_process_is_a_noop():
    from u2tokenizer_amd import replicas
    os.environ.pop("WORLD_SIZE", None)
    os.environ.pop("RANK", None)
    dist, r, w = replicas.init_from_env()
    assert dist is None and (r, w) == (0, 1)
    assert replicas.max_over_ranks(None, 3.5) == 3.5 and replicas.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
