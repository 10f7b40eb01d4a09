"""world_size-2 gloo run of the N>1 path's host logic (bench.py uses the same helpers over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from opticalflowdiffusion_amd import parallel as P
    r, lr, w = P.init(backend="gloo")
    assert (r, w) == (rank, world)
    start, stop = P.shard_batch(32, r, w)
    # every rank "samples" its own chains: disjoint noise streams, no exchange of data
    g = torch.Generator().manual_seed(P.rank_seed(0, r))
    x = torch.randn(stop - start, 2, 4, 4, generator=g)
    P.barrier()
    elapsed = P.max_over_ranks(0.5 + 0.25 * r)
    rate = P.whole_job_rate(5, w, elapsed)
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(w)]
    dist.all_gather(sums, x.double().sum().reshape(1))
    q.put((rank, start, stop, elapsed, rate, [float(s) for s in sums]))
    dist.destroy_process_group()


def test_two_rank_sampling_shards_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, e0, t0, rate0, sums0), (r1, s1, e1, t1, rate1, sums1) = res
    assert (s0, e0, s1, e1) == (0, 16, 16, 32)
    assert t0 == t1 == pytest.approx(0.75)            # MAX over ranks
    assert rate0 == rate1 == pytest.approx(5 * 2 / 0.75)
    assert sums0 == sums1 and sums0[0] != sums0[1]    # different chains per rank
