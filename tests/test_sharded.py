"""CPU: the query-sharded matcher's partition + all-gather logic under gloo, world_size 2 and 3.
The local kernel is injected (the oracle) because no GPU exists here; the GPU path uses the same class."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from visual_slam_amd.sharded import shard_bounds


def test_shard_bounds_cover_queries_exactly():
    for n in (0, 1, 5, 10, 17, 100000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            per = spans[0][2]
            assert all(s[2] == per for s in spans) and per * w >= n
            covered = [i for b, e, _ in spans for i in range(b, e)] if n < 1000 else None
            if covered is not None:
                assert covered == list(range(n))
            assert spans[-1][1] == n or n == 0


def _worker(rank, world, port, nq, nt, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from visual_slam_amd.sharded import ShardedMatcher
    from visual_slam_amd.workloads import match_workload
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local(q, t):
        idx, d = oracle.hamming_knn2(q.numpy(), t.numpy())
        return torch.from_numpy(idx), torch.from_numpy(d)

    q, t = match_workload(nq, nt, n_dup=4, seed=42)
    m = ShardedMatcher(local_knn2=local)
    idx, d = m.knn2(torch.from_numpy(q), torch.from_numpy(t))
    ref_idx, ref_d = oracle.hamming_knn2(q, t)
    ok = np.array_equal(idx.numpy(), ref_idx) and np.array_equal(d.numpy(), ref_d) and idx.shape[0] == nq
    out.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nq,nt", [(2, 101, 64), (2, 64, 200), (3, 50, 40), (2, 1, 10)])
def test_sharded_knn2_equals_single_process(world, nq, nt):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nq, nt, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in results) == list(range(world)) and all(ok for _, ok in results)


def _worker_stream(rank, world, port, out):
    """bench.py's ticket pattern on CPU: submit step k+1, then collect step k (two rotating buffer sets), ragged shards,
    14 steps with different data every step -- first different shapes, then eight steps of one shape, so that each of the
    two buffer sets is overwritten four times while the previous step's ticket is still out."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from visual_slam_amd.sharded import ShardedMatcher, shard_bounds
    from visual_slam_amd.workloads import match_workload
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local(q, t):
        idx, d = oracle.hamming_knn2(q.numpy(), t.numpy())
        return torch.from_numpy(idx), torch.from_numpy(d)

    m = ShardedMatcher(local_knn2=local)
    # repeated shapes re-use the two rotating buffer sets; every step carries different data (seed 200 + k)
    sizes = [(101, 64), (64, 80), (101, 64), (7, 33), (250, 40), (101, 64)] + [(101, 64)] * 8
    work = [match_workload(nq, nt, n_dup=4, seed=200 + k) for k, (nq, nt) in enumerate(sizes)]
    pending, got = [], []
    for q, t in work:
        b, e, _ = shard_bounds(q.shape[0], world, rank)
        ticket = m.submit(torch.from_numpy(q[b:e]), torch.from_numpy(t), q.shape[0])
        if pending:
            i, d = m.collect(pending.pop())
            got.append((i.clone(), d.clone()))
        pending.append(ticket)
    i, d = m.collect(pending.pop())
    got.append((i.clone(), d.clone()))
    ok = len(got) == len(work)
    for (q, t), (i, d) in zip(work, got):
        ri, rd = oracle.hamming_knn2(q, t)
        ok = ok and np.array_equal(i.numpy(), ri) and np.array_equal(d.numpy(), rd)
    out.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_submit_collect_with_rotating_buffers(world):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker_stream, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in results) == list(range(world)) and all(ok for _, ok in results)
