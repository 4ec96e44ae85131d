"""The N > 1 path on CPU: world_size-2 gloo ranks exercise the top-B merge and the Fisher-sum
all-reduce of pool_shard (the only two exchanges of the sharded pool)."""
import os
import socket

import numpy as np
import pytest
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


# ---------------------------------------------------------------------------------------------- sharded AL loop
def _loop_setup():
    from oracle import netspec
    ld = netspec.net_a()
    in_shape = (12, 12, 1)
    pars = netspec.he_init(ld, in_shape, seed=21)
    x = np.random.RandomState(77).randn(240, *in_shape).astype(np.float32)
    return ld, in_shape, pars, x


def _loop_worker(rank, ws, port, q, state_dir=None, rounds=3, ft=False):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    if ws > 1:
        dist.init_process_group('gloo', rank=rank, world_size=ws)
    import nnal_amd  # noqa: F401
    from nnal_amd import al_loop, pool_shard
    from tests.fake_device import FakeModel, FakeSession
    ld, in_shape, pars, x = _loop_setup()
    a, b = pool_shard.shard_bounds(len(x), ws, rank)
    sess = FakeSession()
    model = FakeModel(ld, in_shape, pars, lr=0.01 if ft else None)
    pool = torch.as_tensor(x[a:b].reshape(b - a, -1))
    kw = {}
    if ft:
        from nnal_amd import PW_AL
        kw = dict(labels=(x.reshape(len(x), -1)[:, :7].sum(1) > 0).astype(np.int64), finetune=dict(epochs=2, b=4),
                  state=PW_AL.LoopState(state_dir) if state_dir else None)
    res = al_loop.run_rounds(model, sess, pool, rounds, 20, 6, seed=5, n_global=len(x), **kw)
    out = [{k: r[k] for k in ('queries', 'candidates', 'posts', 'A', 'q', 'pool_left')} for r in res]
    if ft:
        out.append(model.weights())
    q.put((rank, out))
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_loop(ws, *extra):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, ws, port, q) + tuple(extra)) for r in range(ws)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(ws))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_sharded_loop_equals_single_process_bit_for_bit():
    """al_loop.run_rounds over two gloo ranks (each holding its contiguous block of the pool, a fake device backed by
    the oracle) against the same loop in one process: candidates, posteriors, A matrices, query distribution and the
    drawn queries of every round are identical bit for bit, on both ranks."""
    single = _run_loop(1)[0]
    both = _run_loop(2)
    assert len(single) == 3
    taken = []
    for rank in (0, 1):
        for r, (s, d) in enumerate(zip(single, both[rank])):
            for k in ('queries', 'candidates', 'posts', 'A', 'q'):
                np.testing.assert_array_equal(s[k], d[k], err_msg='round %d, %s, rank %d' % (r, k, rank))
            assert s['pool_left'] == d['pool_left']
    for s in single:
        assert 1 <= len(s['queries']) <= 6 and not np.isin(s['queries'], taken).any()
        assert not np.isin(s['candidates'], taken).any()            # queried patches have left the pool
        taken += list(s['queries'])
    # queries span both blocks over the rounds (otherwise the test would not exercise the ownership logic)
    cand = np.concatenate([s['candidates'] for s in single])
    assert (cand < 120).any() and (cand >= 120).any()


def test_sharded_loop_with_finetune_state_and_resume(tmp_path):
    """Config 5 proper: between rounds every rank fine-tunes its replica on all labelled patches (gathered from their
    owners) and rank 0 writes queries/<it>, AL_running_times/dt_<it>, curr_weights_<it>.  Two gloo ranks against one
    process: same queries every round, same weights at the end, bit for bit; the weights really move; and a run that
    is stopped after two rounds and restarted resumes from the files and ends where the uninterrupted one did."""
    d1, d2, d3 = [str(tmp_path / n) for n in ('single', 'double', 'resumed')]
    single = _run_loop(1, d1, 3, True)[0]
    both = _run_loop(2, d2, 3, True)
    for rank in (0, 1):
        for r in range(3):
            for k in ('queries', 'candidates', 'posts', 'A', 'q'):
                np.testing.assert_array_equal(single[r][k], both[rank][r][k], err_msg='round %d %s rank %d' % (r, k, rank))
        for n, wb in single[3].items():
            for a_, b_ in zip(wb, both[rank][3][n]):
                np.testing.assert_array_equal(a_, b_)
    # fine-tuning changes the scores: round 1's posteriors differ from a loop without it
    plain = _run_loop(1)[0]
    np.testing.assert_array_equal(plain[0]['queries'], single[0]['queries'])
    assert not np.array_equal(plain[1]['posts'], single[1]['posts'])
    # state files of the two-rank run: rows [global position, owner rank]
    for it in range(3):
        qm = np.loadtxt(os.path.join(d2, 'queries', '%d' % it), ndmin=2).astype(int)
        np.testing.assert_array_equal(qm[:, 0], single[it]['queries'])
        np.testing.assert_array_equal(qm[:, 1], (qm[:, 0] >= 120).astype(int))
        assert os.path.exists(os.path.join(d2, 'AL_running_times', 'dt_%d' % it))
        assert os.path.exists(os.path.join(d2, 'curr_weights_%d.npz' % (it + 1)))
    # stop after two rounds, restart with rounds = 3: the third round comes out of the files
    _run_loop(1, d3, 2, True)
    resumed = _run_loop(1, d3, 3, True)[0]
    assert len(resumed) == 2                                   # one round executed + the weights
    np.testing.assert_array_equal(resumed[0]['queries'], single[2]['queries'])
    for n, wb in single[3].items():
        for a_, b_ in zip(wb, resumed[1][n]):
            np.testing.assert_array_equal(a_, b_)


# ---------------------------------------------------------------------------------------------- rank launcher
_CHILD = '''
import json, os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
import nnal_amd
from nnal_amd import pool_shard
rank, ws = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=ws)
if len(sys.argv) > 1 and sys.argv[1] == 'fail' and rank == 1:
    sys.exit(7)                                     # dies before the collective: rank 0 would wait for ever
tot = pool_shard.allreduce_sum([float(rank + 1)])
if rank == 0:
    print(json.dumps({'sum': float(tot[0]), 'world': ws, 'local_rank': os.environ['LOCAL_RANK']}))
dist.destroy_process_group()
'''


def test_spawn_ranks_relays_rank0_and_propagates_failure(tmp_path):
    """The parent leg of `python bench.py --gpus N` from a bare shell (pool_shard.spawn_ranks), on CPU."""
    import json
    import sys
    import nnal_amd  # noqa: F401
    from nnal_amd import pool_shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / 'child.py'
    child.write_text(_CHILD % root)
    rc, out = pool_shard.spawn_ranks([sys.executable, str(child)], 2, timeout=120)
    assert rc == 0
    assert json.loads(out.strip().splitlines()[-1]) == {'sum': 3.0, 'world': 2, 'local_rank': '0'}
    rc, out = pool_shard.spawn_ranks([sys.executable, str(child), 'fail'], 2, timeout=120)
    assert rc == 7 and '"sum"' not in out            # rank 0 never got to its JSON line (gloo itself may print a banner)


def test_bench_parent_mode_is_chosen_before_any_gpu_use(monkeypatch):
    """`python bench.py --gpus 2` without WORLD_SIZE must go through spawn_ranks with its own argv and exit with the
    ranks' code, without importing torch.cuda-touching code first."""
    import importlib
    import sys
    import nnal_amd  # noqa: F401
    from nnal_amd import pool_shard
    bench = importlib.import_module('bench')
    seen = {}

    def fake_spawn(argv, n, **kw):
        seen['argv'], seen['n'] = list(argv), n
        return 3, '{"metric": "x"}\n'
    monkeypatch.setattr(pool_shard, 'spawn_ranks', fake_spawn)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '1', '--pool-global', '1000'])
    try:
        bench.main()
        raise AssertionError('bench.main() returned in parent mode')
    except SystemExit as e:
        assert e.code == 3
    assert seen['n'] == 2 and seen['argv'][1].endswith('bench.py') and seen['argv'][2:] == ['--gpus', '2', '--steps', '1', '--pool-global', '1000']


_ATTACH_CHILD = '''
import json, os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
import nnal_amd
from nnal_amd import pool_shard
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['MASTER_PORT'],
                        rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank = dist.get_rank()
mode = sys.argv[1]

class Sess(object):              # the two calls attach_comm makes on a session
    comm_world = 0
    def comm_unique_id(self):
        if mode == 'no_id':
            raise RuntimeError('librccl not loadable')
        return b'u' * 128
    def comm_init(self, uid, r, ws):
        if mode == 'rank1_fails' and r == 1:
            raise RuntimeError('ncclCommInitRank failed')
        self.comm_world = ws

s = Sess()
try:
    pool_shard.attach_comm(s)
    out = 'attached'
except Exception as e:
    out = 'fallback'
# whatever happened, every rank must have taken the same branch - the next collective proves nobody is stuck
agree = pool_shard.max_over_ranks(1.0 if out == 'attached' else 0.0) == (1.0 if out == 'attached' else 0.0)
if rank == 0:
    print(json.dumps({'out': out, 'agree': bool(agree), 'comm_world': s.comm_world}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('mode,want', [('ok', 'attached'), ('rank1_fails', 'fallback'), ('no_id', 'fallback')])
def test_attach_comm_is_all_or_nothing(tmp_path, mode, want):
    """The library's RCCL communicator is attached on every rank or on none: a rank whose init fails (or rank 0 without an
    id) must not leave the others waiting in alq_allreduce_sum while it falls back to torch.distributed."""
    import json
    import sys
    import nnal_amd  # noqa: F401
    from nnal_amd import pool_shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / 'attach_child.py'
    child.write_text(_ATTACH_CHILD % root)
    rc, out = pool_shard.spawn_ranks([sys.executable, str(child), mode], 2, timeout=120)
    assert rc == 0, out
    got = json.loads(out.strip().splitlines()[-1])
    assert got['out'] == want and got['agree']
    assert got['comm_world'] == (2 if want == 'attached' else 0)
