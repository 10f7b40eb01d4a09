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


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from opticalflowdiffusion_amd import parallel as P
    P.init(backend="gloo")
    n = 10_000
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    sync = P.BucketedAllReduce(bucket_bytes=8_000)            # 2000 floats per bucket
    sync.begin(flat)
    # the executor's order: top of the buffer first, neighbouring ops descending, then a far range
    ranges = [(9000, 10000), (8500, 9000), (7000, 8500), (3000, 7000), (2500, 3000), (100, 2500), (0, 100)]
    for b, e in ranges:
        sync.on_range(flat, b, e)
    done = sync.finish(flat)
    others = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(others, mine)
    want = sum(others) / world
    covered = sorted(done)
    q.put((rank, float((flat - want).abs().max()), covered))
    # parameters follow rank 0
    lin = torch.nn.Linear(4, 3)
    P.broadcast_parameters(lin)
    ws = [torch.zeros_like(lin.weight) for _ in range(world)]
    dist.all_gather(ws, lin.weight.data)
    q.put((rank, bool(torch.equal(ws[0], ws[1]))))
    dist.destroy_process_group()


def test_two_rank_bucketed_gradient_allreduce():
    """the data-parallel exchange step (reference: Lightning DDPStrategy, exp_base.py:193-206): ranges
    reported in backward order are coalesced into buckets, every float is averaged exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    avg = [r for r in res if len(r) == 3]
    same = [r for r in res if len(r) == 2]
    for rank, err, covered in avg:
        assert err < 1e-6
        # coalesced: contiguous cover of [0, n) without overlap
        pos = 0
        for b, e in covered:
            assert b == pos
            pos = e
        assert pos == 10_000
        assert len(covered) < 7                       # neighbours were merged into fewer collectives
    assert all(ok for _, ok in same)


def test_bucketed_allreduce_single_process_is_identity():
    from opticalflowdiffusion_amd import parallel as P
    flat = torch.arange(100, dtype=torch.float32)
    sync = P.BucketedAllReduce(bucket_bytes=64)
    sync.begin(flat)
    sync.on_range(flat, 50, 100)
    sync.on_range(flat, 0, 50)
    sync.finish(flat)
    assert torch.equal(flat, torch.arange(100, dtype=torch.float32))


def _bf16_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from opticalflowdiffusion_amd import parallel as P
    P.init(backend="gloo")
    n = 10_000
    g = torch.Generator().manual_seed(200 + rank)
    flat = torch.randn(n, generator=g) * torch.logspace(-6, 2, n)          # gradients spanning eight decades
    mine = flat.clone()
    sync = P.BucketedAllReduce(bucket_bytes=8_000, bucket_dtype="bf16")
    sync.begin(flat)
    for b, e in [(9000, 10000), (8500, 9000), (7000, 8500), (3000, 7000), (2500, 3000), (100, 2500), (0, 100)]:
        sync.on_range(flat, b, e)
    done = sync.finish(flat)
    others = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(others, mine)
    want = sum(o.double() for o in others) / world
    # one bf16 ulp at the magnitude of the larger addend: 2^-7 relative (each input rounded once, the sum once)
    ulp = torch.stack([o.abs() for o in others]).max(dim=0).values.double() * 2.0 ** -7
    q.put((rank, float(((flat.double() - want).abs() / ulp.clamp_min(1e-30)).max()), flat.dtype == torch.float32, sorted(done)[-1][1]))
    dist.destroy_process_group()


def test_two_rank_bf16_gradient_buckets():
    """SURVEY 8e's bf16 gradient buckets (71.5 MB instead of 142.9 MB per step on the wire): cast into the bucket, all-reduce, cast
    back into the flat fp32 buffer; the result stays within one bf16 ulp of the fp32 mean and the buffer stays fp32."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst_ulps, is_f32, end in res:
        assert worst_ulps <= 1.0 and is_f32 and end == 10_000
    from opticalflowdiffusion_amd import parallel as P
    with pytest.raises(ValueError):
        P.BucketedAllReduce(bucket_dtype="fp8")
