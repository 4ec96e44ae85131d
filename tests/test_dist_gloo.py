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


def _worker8(rank, ws, port, n, B, q):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    import nnal_amd  # noqa: F401
    from nnal_amd import pool_shard
    from tests.fake_device import FakeSession
    sess = FakeSession()
    rs = np.random.RandomState(321)
    key = np.abs(rs.randint(0, 40, size=n) / 40. - .5)            # heavy ties, also across ranks
    A = rs.randn(n, 2, 2)
    a, b = pool_shard.shard_bounds(n, ws, rank)
    lk, lg = key[a:b], np.arange(a, b)
    o = np.lexsort((lg, lk))[:B]                                  # what the device filter hands over: ascending, tie-ordered
    sel_dev = pool_shard.merge_topB_device(sess, torch.as_tensor(lk[o]), torch.as_tensor(lg[o]), min(B, n)).numpy()
    # with the pool size the number of real candidates follows on the host (no read-back of the gathered indices)
    sel_dev2 = pool_shard.merge_topB_device(sess, torch.as_tensor(lk[o]), torch.as_tensor(lg[o]), min(B, n), n_global=n).numpy()
    assert np.array_equal(sel_dev, sel_dev2)
    sel_host = pool_shard.merge_topB(lk, lg, min(B, n))
    Asum = pool_shard.allreduce_sum_device(torch.as_tensor(A[a:b].sum(0)), sess).numpy()
    rows = pool_shard.allgather_rows(n, np.arange(a, b), A[a:b])
    blocks = pool_shard.work_block(n)
    pool_shard.barrier()
    q.put((rank, sel_dev, sel_host, Asum, rows, blocks, (a, b)))
    dist.destroy_process_group()


@pytest.mark.parametrize('n,B', [(43, 16), (5, 16), (41, 41)])
def test_world8_ragged_and_empty_shards(n, B):
    """Eight ranks, a pool that does not divide by eight (43 -> blocks of 6, the last holds 1; 41 -> the last rank is
    EMPTY; 5 -> three ranks are empty and B exceeds the pool): the device-tensor merge (merge_topB_device), the host merge,
    the Fisher all-reduce and the row all-gather agree with the single-process answer on every rank."""
    ws = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, ws, port, n, B, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(ws)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rs = np.random.RandomState(321)
    key = np.abs(rs.randint(0, 40, size=n) / 40. - .5)
    A = rs.randn(n, 2, 2)
    want = np.lexsort((np.arange(n), key))[:min(B, n)]
    covered = np.zeros(n, int)
    for rank, sel_dev, sel_host, Asum, rows, blocks, (a, b) in res:
        np.testing.assert_array_equal(sel_dev, want)
        np.testing.assert_array_equal(sel_host, want)
        np.testing.assert_allclose(Asum, A.sum(0), rtol=1e-12, atol=1e-14)
        np.testing.assert_array_equal(rows, A)
        assert blocks == (a, b)
        covered[a:b] += 1
    assert (covered == 1).all()                                   # the blocks tile the pool exactly once
    assert any(a == b for *_, (a, b) in res) or n % ws == 0 or n == 43


def _strong_worker(rank, ws, port, n, B, q):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    if ws > 1:
        dist.init_process_group('gloo', rank=rank, world_size=ws)
    import nnal_amd  # noqa: F401
    from nnal_amd import pool_shard
    from tests.fake_device import FakeModel, FakeSession
    ld, in_shape, pars, _ = _loop_setup()
    x = np.random.RandomState(78).randn(n, *in_shape).astype(np.float32)
    sess = FakeSession()
    model = FakeModel(ld, in_shape, pars)
    a, b = pool_shard.shard_bounds(n, ws, rank)
    out = pool_shard.score_pool(model, sess, torch.as_tensor(x[a:b].reshape(b - a, -1)), n, B, 1e-3, want=('p1', 'A', 'Asum'))
    q.put((rank, out['sel'].numpy(), out['Asum'].numpy(), out['offset'], (a, b)))
    if ws > 1:
        pool_shard.barrier()
        dist.destroy_process_group()


def test_score_pool_strong_mode_world8_ragged_last_block():
    """bench.py's strong mode (--pool-global: ONE pool in contiguous blocks of ceil(G / N), configs[3]) at world 8 with a ragged
    last block (301 patches -> seven blocks of 38 and one of 35): `pool_shard.score_pool` - what one bench step runs - gives on
    every rank the single-process selection bit for bit (ties -> lower global index) and the single-process Fisher sum (to the
    last bits of another summation order; per-patch A matrices are pure functions of the patch)."""
    n, B = 301, 64
    ctx = mp.get_context('spawn')

    def run(ws):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_strong_worker, args=(r, ws, port, n, B, q)) for r in range(ws)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in range(ws)]
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        return sorted(res, key=lambda t: t[0])
    one = run(1)[0]
    many = run(8)
    assert many[-1][4] == (266, 301) and many[0][4] == (0, 38)
    for rank, sel, Asum, off, (a, b) in many:
        np.testing.assert_array_equal(sel, one[1])
        np.testing.assert_allclose(Asum, one[2], rtol=1e-12, atol=1e-15)
        assert off == a


# ---------------------------------------------------------------------------------------------- sharded AL loop
def _loop_setup():
    from oracle import netspec
    ld = netspec.net_a()
    in_shape = (12, 12, 1)
    pars = netspec.he_init(ld, in_shape, seed=21)
    x = np.random.RandomState(77).randn(240, *in_shape).astype(np.float32)
    return ld, in_shape, pars, x


def _loop_worker(rank, ws, port, q, state_dir=None, rounds=3, ft=False, n_pool=None):
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
    if n_pool:
        x = x[:n_pool]
    a, b = pool_shard.shard_bounds(len(x), ws, rank)
    sess = FakeSession()
    model = FakeModel(ld, in_shape, pars, lr=0.01 if ft else None)
    pool = torch.as_tensor(x[a:b].reshape(b - a, int(np.prod(x.shape[1:]))))
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


def test_sharded_loop_world8_with_an_empty_shard():
    """The same loop over EIGHT ranks on a 41-patch pool (blocks of 6: rank 6 holds 5, rank 7 nothing) with fine-tuning:
    every rank reports the single-process rounds and ends with the single-process weights, bit for bit."""
    single = _run_loop(1, None, 2, True, 41)[0]
    eight = _run_loop(8, None, 2, True, 41)
    for rank in range(8):
        for r in range(2):
            for k in ('queries', 'candidates', 'posts', 'A', 'q'):
                np.testing.assert_array_equal(single[r][k], eight[rank][r][k], err_msg='round %d %s rank %d' % (r, k, rank))
        for n, wb in single[2].items():
            for a_, b_ in zip(wb, eight[rank][2][n]):
                np.testing.assert_array_equal(a_, b_)


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

class Sess(object):              # the calls attach_comm makes on a session
    comm_world = 0
    inits = 0
    destroyed = 0
    def comm_unique_id(self):
        if mode == 'no_id' or (mode == 'rank1_no_rccl' and rank == 1):
            raise RuntimeError('librccl not loadable')
        return b'u' * 128
    def comm_init(self, uid, r, ws):
        self.inits += 1
        if mode == 'rank1_fails' and r == 1:
            raise RuntimeError('ncclCommInitRank failed')
        self.comm_world = ws
    def comm_destroy(self):
        self.destroyed += 1
        self.comm_world = 0

s = Sess()
try:
    pool_shard.attach_comm(s)
    out = 'attached'
except Exception as e:
    out = 'fallback'
# whatever happened, every rank must have taken the same branch - the next collective proves nobody is stuck
agree = pool_shard.max_over_ranks(1.0 if out == 'attached' else 0.0) == (1.0 if out == 'attached' else 0.0)
inits = pool_shard.max_over_ranks(float(s.inits))
if rank == 0:
    print(json.dumps({'out': out, 'agree': bool(agree), 'comm_world': s.comm_world, 'max_inits': inits, 'destroyed': s.destroyed}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('mode,want', [('ok', 'attached'), ('rank1_fails', 'fallback'), ('no_id', 'fallback'), ('rank1_no_rccl', 'fallback')])
def test_attach_comm_is_all_or_nothing(tmp_path, mode, want):
    """The library's RCCL communicator is attached on every rank or on none: a rank whose init fails must not leave the
    others waiting in alq_allreduce_sum while it falls back to torch.distributed, and the ranks that did get a
    communicator give it back (alq_comm_destroy).  A rank that cannot reach RCCL at all is found by a vote BEFORE the
    collective init, so no rank enters ncclCommInitRank to wait for it."""
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
    if mode in ('no_id', 'rank1_no_rccl'):
        assert got['max_inits'] == 0                      # nobody entered the collective init
    if mode == 'rank1_fails':
        assert got['destroyed'] == 1                      # rank 0 had a communicator and gave it back


# ---------------------------------------------------------------------------------------------- volume-level experiment
def _write_subjects(root, seed=4100):
    """Two synthetic subjects (two modalities + a mask with NaN voxels) as NRRD files."""
    import nnal_amd  # noqa: F401
    from nnal_amd import nrrd_io
    rs = np.random.RandomState(seed)
    paths = []
    for s_, shp in enumerate([(14, 12, 6), (12, 15, 5)]):
        sub = []
        for j in range(2):
            p = os.path.join(root, 'sub%d_mod%d.nrrd' % (s_, j))
            nrrd_io.write(p, rs.randn(*shp) * (1. + j) + 0.2 * s_)
            sub.append(p)
        mask = rs.randint(0, 2, size=shp).astype(np.float64)
        mask[rs.rand(*shp) < 0.1] = np.nan
        p = os.path.join(root, 'sub%d_mask.nrrd' % s_)
        nrrd_io.write(p, mask)
        sub.append(p)
        paths.append(sub)
    return paths


VOL_PARS = dict(grid_spacing=2, patch_shape=(5, 5, 3), model_name='NET-A', dropout_rate=1., learning_rate=0.02, grad_layers=[],
                train_layers=[], optimizer_name='SGD', init_weights_path='init', k=5, B=30, lambda_=0., ntb=40, b=4, epochs=2)


def _experiment_worker(rank, ws, port, q, root, data, rounds):      # rounds = max_queries of run_method
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    if ws > 1:
        dist.init_process_group('gloo', rank=rank, world_size=ws)
    import nnal_amd  # noqa: F401
    from nnal_amd import PW_AL, patch_utils
    from oracle import alpath, netspec
    from tests.fake_device import FakeModel, FakeSession, FakeVolumes
    patch_utils.DeviceVolumes = FakeVolumes                       # the gather on the CPU (no GPU in this test)
    patch_utils.get_patches_multimg = alpath.get_patches_multimg
    # every rank constructs the experiment at once, as a one-process-per-GPU launch does: rank 0 writes parameters / paths /
    # stats, the others wait inside the constructor and read the finished files back (round-3 advisor finding: the
    # constructors used to race)
    expr = PW_AL.Experiment_MultiImg(root, VOL_PARS, _subject_paths(data))
    sess = FakeSession()

    def factory(e, in_shape, s):
        ld = netspec.net_a()
        m = FakeModel(ld, in_shape, netspec.he_init(ld, in_shape, seed=61, bias_std=0.05), lr=e.pars['learning_rate'])
        s.model = m
        return m
    expr.model_factory = factory
    expr.add_method('fi')                                                     # (rank-safe as well)
    np.random.seed(17)
    log = expr.run_method('fi', rounds, sess=sess)
    q.put((rank, [l['Q_mat'] for l in log], expr.model.weights()))
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


def _subject_paths(data):
    return [[os.path.join(data, 'sub%d_%s.nrrd' % (s_, t)) for t in ('mod0', 'mod1', 'mask')] for s_ in range(2)]


def _run_experiment(ws, root, data, rounds):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_experiment_worker, args=(r, ws, port, q, root, data, rounds)) for r in range(ws)]
    for p in procs:
        p.start()
    res = {r: (Q, w) for r, Q, w in (q.get(timeout=600) for _ in range(ws))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_volume_level_experiment_world2_equals_single_process(tmp_path):
    """PW_AL.Experiment_MultiImg.run_method (the reference's config-5 control flow, PW_AL.py:690-898) over synthetic NRRD
    subjects: gen_multimg_inds -> query_multimg('fi') -> queries/<iter> -> finetune_multimg -> curr_weights_<iter> until
    six voxels are queried (the draws of a round collapse onto the support of the SDP's optimum, np.unique in
    sample_query_dstr, so a round yields 1 .. k queries).  Two gloo ranks (the query's device work split by contiguous
    blocks of the pool and of the candidate list) against one process: the query files, every round's Q_mat and the
    final weights agree bit for bit; a run stopped after three queries and resumed continues with the rounds of the
    uninterrupted run."""
    data = str(tmp_path / 'data')
    os.makedirs(data)
    _write_subjects(data)
    one = _run_experiment(1, str(tmp_path / 'e1'), data, 6)
    two = _run_experiment(2, str(tmp_path / 'e2'), data, 6)
    Q1, w1 = one[0]
    assert len(Q1) >= 2 and all(len(Q) > 0 for Q in Q1) and sum(len(Q) for Q in Q1) >= 6
    for r in (0, 1):
        Qr, wr = two[r]
        for a, b in zip(Q1, Qr):
            np.testing.assert_array_equal(a, b)
        for n in w1:
            for a, b in zip(w1[n], wr[n]):
                np.testing.assert_array_equal(a, b)
    for it in range(len(Q1)):
        f1 = np.loadtxt(os.path.join(str(tmp_path / 'e1'), 'fi', 'queries', '%d' % it), ndmin=2)
        f2 = np.loadtxt(os.path.join(str(tmp_path / 'e2'), 'fi', 'queries', '%d' % it), ndmin=2)
        np.testing.assert_array_equal(f1, f2)
        np.testing.assert_array_equal(f1.astype(np.int64), Q1[it])
        assert os.path.exists(os.path.join(str(tmp_path / 'e2'), 'fi', 'curr_weights_%d.npz' % (it + 1)))
    # no queried voxel is queried twice, and every query is a grid voxel with a non-NaN mask
    allq = np.concatenate(Q1)
    assert len(np.unique(allq, axis=0)) == len(allq)
    # resume: stop after the rounds that reach three queries, run again up to six in total
    part = _run_experiment(1, str(tmp_path / 'e3'), data, 3)[0][0]
    n_part = len(part)
    for a, b in zip(Q1, part):
        np.testing.assert_array_equal(a, b)
    rest, w3 = _run_experiment(1, str(tmp_path / 'e3'), data, 2)[0]
    # the resumed run continues the numbering, starts from the saved weights' pool state and never re-queries a voxel
    # (its RNG stream restarts, as in the reference, so its draws are its own)
    assert sorted(os.listdir(os.path.join(str(tmp_path / 'e3'), 'fi', 'queries')), key=int) == [str(i) for i in range(n_part + len(rest))]
    both = np.concatenate(part + rest)
    assert len(np.unique(both, axis=0)) == len(both)
