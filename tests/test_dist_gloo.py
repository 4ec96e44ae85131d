"""The N > 1 path on CPU: world_size-2 gloo ranks exercise the top-B merge and the Fisher-sum
all-reduce of pool_shard (the only two exchanges of the sharded pool)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n, B, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import nnal_amd  # noqa: F401
    from nnal_amd import pool_shard
    rs = np.random.RandomState(123)
    p = (rs.randint(0, 500, size=n) / 500.)          # ties across ranks
    key = np.abs(p - .5)
    A = rs.randn(n, 3, 3)
    a, b = pool_shard.shard_bounds(n, ws, rank)
    lk, lg = key[a:b], np.arange(a, b)
    sel = pool_shard.merge_topB(lk, lg, B)
    Asum = pool_shard.allreduce_sum(A[a:b].sum(0))
    mx = pool_shard.max_over_ranks(float(rank + 1))
    pool_shard.barrier()
    q.put((rank, sel, Asum, mx))
    dist.destroy_process_group()


def test_world2_merge_and_allreduce():
    n, B, ws = 1001, 64, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, n, B, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rs = np.random.RandomState(123)
    pv = (rs.randint(0, 500, size=n) / 500.)
    key = np.abs(pv - .5)
    A = rs.randn(n, 3, 3)
    want = np.lexsort((np.arange(n), key))[:B]
    for rank, sel, Asum, mx in res:
        np.testing.assert_array_equal(sel, want)            # identical on every rank, ties -> lower index
        np.testing.assert_allclose(Asum, A.sum(0), rtol=1e-12)
        assert mx == 2.0
