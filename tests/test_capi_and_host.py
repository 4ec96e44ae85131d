"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/alq.h
declares (no compute calls without a GPU), and the host logic of the Python mirror."""
import os
import re
import subprocess

import numpy as np
import pytest

import nnal_amd  # noqa: F401
from nnal_amd import _lib, device, NNAL_tools, patch_utils, pool_shard, PW_NNAL
from oracle import netspec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'alq.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(alq_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    L = _lib.lib()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), 'libalq.so does not export %s' % n
    assert sorted(names) == _lib.exported_names()      # the ctypes table covers the header exactly
    nm = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    for n in names:
        assert re.search(r'\bT %s\b' % n, nm), n
    assert L.alq_version() == 1
    assert L.alq_prof_num_classes() >= 2 and L.alq_prof_class_name(0) == b'igemm_fwd'
    assert L.alq_topk_work_bytes(5) == 2048 * 16 and L.alq_topk_work_bytes(2049) == 4096 * 16


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.AlqError):
        device.DeviceSession(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'nn-active-learning_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f
                assert '/root/reference' not in txt, f


def test_translate_layers_both_schemas():
    a = device.translate_layers(netspec.net_b(), (25, 25, 2))
    assert [d['type'] for d in a] == [0, 0, 2, 0, 0, 2, 3, 3, 3]
    assert a[0]['k'] == [1, 5, 5] and a[2]['k'] == [1, 2, 2] and a[2]['s'] == [1, 2, 2]
    assert [d['relu'] for d in a] == [1, 1, 0, 1, 1, 0, 1, 1, 0]          # no ReLU on the last fc
    shapes = device.tf_param_shapes(a, (25, 25, 2))
    assert shapes[4] == ('fc1', (4096, 7 * 7 * 96), (4096, 1))
    assert sum(int(np.prod(w)) + int(np.prod(b)) for _, w, b in shapes) == 36137082
    ld, sk = netspec.net_c()
    c = device.translate_layers(ld, (32, 32, 32, 1), sk)
    assert c[6]['skip_src'] == 2 and c[8]['skip_src'] == 0 and c[5]['type'] == 1 and c[5]['relu'] == 0
    shapes = device.tf_param_shapes(c, (32, 32, 32, 1))
    assert sum(int(np.prod(w)) + int(np.prod(b)) for _, w, b in shapes) == 576450
    # same shapes as the oracle's independent derivation
    assert [(n, tuple(w), tuple(b)) for n, w, b in netspec.param_shapes(ld, (32, 32, 32, 1), sk)] == \
           [(n, tuple(w), tuple(b)) for n, w, b in shapes]
    with pytest.raises(NotImplementedError):
        device.translate_layers(ld, (32, 32, 32, 1), [[0, [8], 'sum']])


def test_host_functions_vs_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, 'host_layers.npz'))
    np.testing.assert_array_equal(NNAL_tools.shrink_gradient([g['shrink_in_%d' % i] for i in range(6)], 'sum'),
                                  g['shrink_out'])
    for i, l in enumerate(patch_utils.global2local_inds(g['g2l_inds'], list(g['g2l_sizes']))):
        np.testing.assert_array_equal(l, g['g2l_out_%d' % i])
    np.testing.assert_array_equal(PW_NNAL.binary_uncertainty_filter(g['buf_posts'], 37), g['buf_out'])
    a = g['ent_in'].copy()
    np.testing.assert_array_equal(NNAL_tools.compute_entropy(a), g['ent_out'])
    np.testing.assert_array_equal(a, g['ent_in_after'])
    b = g['ent_in'].copy()
    np.testing.assert_array_equal(NNAL_tools.uncertainty_filtering(b, 11), g['uf_out'])
    np.testing.assert_array_equal(b, g['uf_in_after'])
    np.random.seed(77)
    np.testing.assert_array_equal(NNAL_tools.sample_query_dstr(g['sq_q'].copy(), 25, True), g['sq_out'])
    with pytest.raises(ValueError):
        NNAL_tools.SDP_query_distribution([np.eye(2)], 0.1, [], 3)      # feature-regularised form needs X_pool [d, n]


def test_sdp_query_distribution_is_the_a_optimal_design():
    """min tr((sum q_i A_i)^-1) over the simplex: KKT conditions at the returned q, and agreement with
    a generic constrained optimiser on a small instance (the reference hands this SDP to cvxopt)."""
    from scipy.optimize import minimize
    rs = np.random.RandomState(4)
    n, L = 12, 3
    G = rs.randn(n, L)
    A = np.stack([np.outer(g, g) * rs.rand() + 1e-3 * np.eye(L) for g in G])
    soln = NNAL_tools.SDP_query_distribution(list(A), 0., [], 5)
    q, t = soln['x'][:n], soln['x'][n:]
    assert soln['status'].startswith('optimal') and abs(q.sum() - 1) < 1e-12 and q.min() >= 0
    M = np.tensordot(q, A, axes=(0, 0))
    Minv = np.linalg.inv(M)
    np.testing.assert_allclose(t, np.diag(Minv), rtol=1e-12)
    d = np.tensordot(A, Minv @ Minv, axes=([1, 2], [0, 1]))
    assert d.max() <= np.trace(Minv) * (1 + 1e-6)                          # dual feasibility
    np.testing.assert_allclose(d[q > 1e-6], np.trace(Minv), rtol=1e-4)       # complementary slackness

    def f(z):
        qq = np.abs(z) / np.abs(z).sum()
        return np.trace(np.linalg.inv(np.tensordot(qq, A, axes=(0, 0))))
    best = min(minimize(f, rs.rand(n), method='Nelder-Mead', options={'maxiter': 20000, 'xatol': 1e-10, 'fatol': 1e-12}).fun
               for _ in range(3))
    assert soln['primal objective'] <= best * (1 + 1e-6)
    qm, obj = NNAL_tools.solve_FIAL_SDP(list(A))
    np.testing.assert_allclose(obj, soln['primal objective'], rtol=1e-12)


@pytest.mark.parametrize('n,L', [(1, 2), (3, 8), (60, 4), (700, 8)])
def test_sdp_newton_and_multiplicative_solvers_agree(n, L):
    """The two own solvers of the A-optimal design (log-barrier Newton on the diagonal + low-rank Hessian, and
    the first-order multiplicative algorithm) reach the same optimum: same objective within the stopping
    tolerance, same information matrix M(q) (unique at the optimum), each with its optimality certificate."""
    rs = np.random.RandomState(100 + n)
    G0, G1, p = rs.randn(n, L), rs.randn(n, L), rs.rand(n)
    A = np.stack([(1 - pp) * np.outer(a, a) + pp * np.outer(b, b) + 1e-3 * np.eye(L) for a, b, pp in zip(G0, G1, p)])
    sn = NNAL_tools.SDP_query_distribution(A, 0., [], 5, tol=1e-8)
    sm = NNAL_tools.SDP_query_distribution(A, 0., [], 5, tol=1e-8, max_iter=200000, method='multiplicative')
    for s_ in (sn, sm):
        q = s_['x'][:n]
        assert s_['status'].startswith('optimal') and s_['gap'] <= 1e-8 and abs(q.sum() - 1) < 1e-12 and q.min() >= 0
    assert sn['iterations'] < 200
    np.testing.assert_allclose(sn['primal objective'], sm['primal objective'], rtol=3e-8)
    Mn = np.tensordot(sn['x'][:n], A, axes=(0, 0))
    Mm = np.tensordot(sm['x'][:n], A, axes=(0, 0))
    np.testing.assert_allclose(Mn, Mm, rtol=0, atol=2e-3 * np.abs(Mm).max())


def test_shard_bounds_cover_the_pool():
    for n in (0, 1, 7, 8, 100000, 1000003):
        for R in (1, 2, 3, 8):
            b = [pool_shard.shard_bounds(n, R, r) for r in range(R)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(R - 1))
            assert all(0 <= x[1] - x[0] <= -(-n // R) for x in b)


def test_merge_topB_single_process():
    k = np.array([.3, .1, .1, .2])
    g = np.array([10, 7, 3, 5])
    np.testing.assert_array_equal(pool_shard.merge_topB(k, g, 3), [3, 7, 5])
    np.testing.assert_array_equal(pool_shard.allreduce_sum(np.eye(2)), np.eye(2))


def test_product_netspec_matches_the_oracle_copy():
    """bench.py and al_loop build their models from the product's netspec; the oracle keeps its own copy."""
    import nnal_amd  # noqa: F401
    from nnal_amd import netspec as prod
    from oracle import netspec as orc
    for mk, shape in (('net_a', (32, 32, 1)), ('net_b_small', (25, 25, 2)), ('net_c_2d', (16, 16, 1))):
        a, b = getattr(prod, mk)(), getattr(orc, mk)()
        sk = ()
        if isinstance(a, tuple):
            (a, sk), (b, _) = a, b
        assert list(a.items()) == list(b.items())
        pa, pb = prod.he_init(a, shape, seed=5, skips=sk), orc.he_init(b, shape, seed=5, skips=sk)
        assert list(pa) == list(pb)
        for k in pa:
            np.testing.assert_array_equal(pa[k][0], pb[k][0])
            np.testing.assert_array_equal(pa[k][1], pb[k][1])
    (la, ska), (lb, skb) = prod.net_c(), orc.net_c()
    assert list(la.items()) == list(lb.items()) and ska == skb


def test_topk_merge_c_abi():
    """alq_topk_merge: ascending key, ties -> lower global index, padding dropped (host function, no GPU)."""
    import ctypes as C
    import nnal_amd  # noqa: F401
    from nnal_amd._lib import lib, check
    rs = np.random.RandomState(3)
    k = np.round(rs.rand(500), 2)                   # many ties
    g = rs.permutation(5000)[:500].astype(np.int64)
    g[::7] = -1                                     # padding
    out = np.empty(64, np.int64)
    n_out = C.c_int64(0)
    check(lib().alq_topk_merge(k.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), 500, 64,
                               out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    keep = g >= 0
    want = g[keep][np.lexsort((g[keep], k[keep]))][:64]
    assert n_out.value == 64
    np.testing.assert_array_equal(out, want)


def test_weight_files_npz_twin_and_hdf5_when_available(tmp_path):
    """weights_io: the reference keeps weights in HDF5 (groups per layer, datasets Weight / Bias in TF layouts,
    NN.py:379-419; NN_extended.py:670-693 names the datasets after the variables).  The .npz twin round-trips everywhere; the
    HDF5 form round-trips when h5py is importable and fails with a clear ImportError (not a guess) when it is not."""
    import nnal_amd  # noqa: F401
    from nnal_amd import weights_io
    rs = np.random.RandomState(3)
    vd = {'conv1': (rs.randn(3, 3, 1, 4).astype(np.float32), rs.randn(4).astype(np.float32)),
          'fc1': (rs.randn(2, 36).astype(np.float32), rs.randn(2, 1).astype(np.float32))}
    p = str(tmp_path / 'w.npz')
    weights_io.write_weights(p, vd)
    got = weights_io.read_weights(p, list(vd))
    for n in vd:
        np.testing.assert_array_equal(got[n][0], vd[n][0])
        np.testing.assert_array_equal(got[n][1], vd[n][1])
    h5 = str(tmp_path / 'curr_weights.h5')
    if not weights_io.have_h5py():
        with pytest.raises(ImportError, match='h5py'):
            weights_io.write_weights(h5, vd)
        with pytest.raises(ImportError, match='h5py'):
            weights_io.read_weights(h5, list(vd))
        return
    import h5py
    weights_io.write_weights(h5, vd)
    with h5py.File(h5, 'r') as f:       # the reference's own reader: f[layer]['Weight'] / ['Bias'] (NN.py:409-419)
        for n in vd:
            np.testing.assert_array_equal(np.array(f[n]['Weight']), vd[n][0])
            np.testing.assert_array_equal(np.array(f[n]['Bias']), vd[n][1])
    got = weights_io.read_weights(h5, list(vd))
    for n in vd:
        np.testing.assert_array_equal(got[n][0], vd[n][0])
    # NN_extended's naming: datasets named after the variables
    with h5py.File(str(tmp_path / 'ext.h5'), 'w') as f:
        g = f.create_group('conv1')
        g.create_dataset('Weight_1', data=vd['conv1'][0])
        g.create_dataset('Bias_1', data=vd['conv1'][1])
    got = weights_io.read_weights(str(tmp_path / 'ext.h5'), ['conv1'])
    np.testing.assert_array_equal(got['conv1'][0], vd['conv1'][0])


def test_run_method_bookkeeping_against_the_reference_run(golden_dir, tmp_path, monkeypatch):
    """PW_AL.Experiment_MultiImg.run_method's bookkeeping against a run of the reference's own lines (PW_AL.py:690-898;
    fixture r4_run_method.npz, made by tests/golden/make_golden_r4.py with stand-ins for I/O, TensorFlow and the query):
    grid pool, volume statistics, the pool each query sees, the training lists each fine-tune is handed, the `queries/<iter>`
    files byte for byte, and the resume path (a second run_method on the same directory) - all bit-exact."""
    import copy
    from nnal_amd import PW_AL, nrrd_io
    sys_path = os.path.join(os.path.dirname(__file__), 'golden')
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_r4', os.path.join(sys_path, 'make_golden_r4.py'))
    gen = importlib.util.module_from_spec(spec)
    import sys
    sys.path.insert(0, sys_path)
    try:
        spec.loader.exec_module(gen)              # data + the seeded picker only; main() (the reference run) is not called
    finally:
        sys.path.remove(sys_path)
    g = np.load(os.path.join(golden_dir, 'r4_run_method.npz'))
    table, paths = gen.subjects(int(g['subject_seed']))
    data = tmp_path / 'data'
    data.mkdir()
    here = []
    for sub in paths:
        row = []
        for p in sub:
            q = str(data / os.path.basename(p))
            nrrd_io.write(q, table[p])
            row.append(q)
        here.append(row)
    rec = {'pools_seen': [], 'train_seen': []}
    pick = gen.picker(int(g['pick_seed']))

    def fake_query(expr, model, sess, all_padded_imgs, pool_inds, labeled_inds, method_name):
        rec['pools_seen'].append(copy.deepcopy(pool_inds))
        return pick(pool_inds)

    def fake_finetune(expr, model, sess, all_padded_imgs, training_inds):
        rec['train_seen'].append(copy.deepcopy(training_inds))

    class Model(object):
        def add_assign_ops(self):
            pass

        def perform_assign_ops(self, path, sess):
            pass

        def save_weights(self, path):
            pass
    monkeypatch.setattr(PW_NNAL, 'query_multimg', fake_query)
    monkeypatch.setattr(PW_AL, 'finetune_multimg', fake_finetune)
    root = str(tmp_path / 'expr')
    expr = PW_AL.Experiment_MultiImg(root, dict(gen.PARS), here)
    expr.model_factory = lambda e, shp, s: Model()
    expr.add_method('fi')
    per_iter = int(np.sum(g['picks']))
    np.testing.assert_array_equal(expr.train_stats, g['train_stats'])
    expr.run_method('fi', 3 * per_iter, sess=object())
    expr2 = PW_AL.Experiment_MultiImg(root)
    expr2.model_factory = lambda e, shp, s: Model()
    np.testing.assert_array_equal(expr2.train_stats, g['train_stats_reloaded'])
    expr2.run_method('fi', 2 * per_iter, sess=object())
    qdir = os.path.join(root, 'fi', 'queries')
    assert sorted(os.listdir(qdir), key=int) == ['0', '1', '2', '3', '4']
    assert sorted(os.listdir(os.path.join(root, 'fi', 'AL_running_times'))) == ['dt_%d' % i for i in range(5)]
    for it in range(5):
        with open(os.path.join(qdir, '%d' % it), 'rb') as f:
            assert f.read() == g['queries_text_%d' % it].tobytes(), 'queries/%d differs from the reference run' % it
        for s_ in range(2):
            np.testing.assert_array_equal(np.asarray(rec['pools_seen'][it][s_], dtype=np.int64), g['pool_seen_%d_%d' % (it, s_)])
            mine = np.asarray(rec['train_seen'][it][s_], dtype=np.int64)
            if it >= 3:
                # resumed: the reference concatenates the query files in os.listdir order (PW_AL.py:726-734, the file system's
                # choice), this repo in iteration order; same per-file blocks, so put mine in the order the reference run saw
                k = int(g['picks'][s_])
                blocks = [mine[j * k:(j + 1) * k] for j in range(it + 1)]
                mine = np.concatenate([blocks[j] for j in g['resume_listdir_order']] + blocks[3:])
            np.testing.assert_array_equal(mine, g['train_seen_%d_%d' % (it, s_)])


def test_store_hazard_gate_flags_an_early_rewrite_of_store_data(tmp_path):
    """csrc/build.sh runs tools/isa_store_hazard.py over the sweep kernels' device assembly: a 16-byte buffer store with a
    REGISTER soffset (the compiler inserts no wait states for it, alq_internal.h ALQ_STORE_HOLD) must keep its data registers
    untouched for 2 wait states.  The gate itself, on hand-written assembly: an early VALU rewrite fails, the same code with
    the hold passes, a store with an immediate soffset is the compiler's business, a rewrite on the taken arm of a branch is
    found, a load into the data registers (written on return, much later) is not counted."""
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'isa_store_hazard.py')

    def run(body):
        f = tmp_path / 'k.s'
        f.write_text('\t.text\nkern:\n' + body + '\ts_endpgm\n')
        return subprocess.run([sys.executable, tool, '--min', '2', str(f)], capture_output=True, text=True)

    bad = '\tbuffer_store_dwordx4 v[4:7], v0, s[8:11], s3 offen\n\tv_add_f32_e32 v5, v1, v2\n'
    r = run(bad)
    assert r.returncode == 1 and 'HAZARD (0 wait states' in r.stderr
    r = run('\tbuffer_store_dwordx4 v[4:7], v0, s[8:11], s3 offen\n\ts_nop 1\n\tv_add_f32_e32 v5, v1, v2\n')
    assert r.returncode == 0, r.stderr
    r = run('\tbuffer_store_dwordx4 v[4:7], v0, s[8:11], s3 offen\n\ts_mov_b32 s0, 0\n\tv_mfma_f32_16x16x32_f16 v[4:7], v[10:13], v[14:17], v[4:7]\n')
    assert r.returncode == 1 and 'HAZARD (1 wait states' in r.stderr
    assert run('\tbuffer_store_dwordx4 v[4:7], v0, s[8:11], 0 offen\n\tv_add_f32_e32 v5, v1, v2\n').returncode == 0
    assert run('\tbuffer_store_dwordx4 v[4:7], v0, s[8:11], s3 offen\n\tbuffer_load_dwordx4 v[4:7], v0, s[8:11], s3 offen\n'
               '\ts_nop 0\n\tv_add_f32_e32 v9, v1, v2\n').returncode == 0
    r = run('\tbuffer_store_dwordx4 v[4:7], v0, s[8:11], s3 offen\n\ts_cbranch_scc1 .LBB0_2\n\ts_nop 3\n\tv_mov_b32_e32 v4, 0\n'
            '.LBB0_2:\n\tv_mov_b32_e32 v6, 0\n')
    assert r.returncode == 1 and 'HAZARD (1 wait states' in r.stderr
    # the product build's own report: every sweep-kernel store held for at least 2 wait states
    rep = os.path.join(os.path.dirname(_lib.LIB_PATH), 'csrc', 'build', 'store_hazard_report.txt')
    if os.path.exists(rep):
        for ln in open(rep):
            m = re.search(r'(\d+) 12/16-byte stores.*rewritten: (\w+)', ln)
            assert m and (int(m.group(1)) == 0 or int(m.group(2)) >= 2), ln


def test_ref64_host_side_scores_and_flip_records():
    """nnal_amd/ref64.py, the host half of the fp64 device reference: `scores` forms p1, g0, g1, A from logits and layer sums exactly as
    gen_A_matrices does (tests/factored_ref.fisher_from_unit restates PW_NNAL.py:770-814), saturation branches included, and the
    flip record is the 24-byte alq_flip_t of include/alq.h."""
    from nnal_amd import ref64
    from tests import factored_ref
    assert ref64.FLIP_DTYPE.itemsize == 24 and ref64.FLIP_DTYPE.names == ('layer', 'pad', 'idx', 'delta')
    hdr = open(os.path.join(os.path.dirname(_lib.LIB_PATH), '..', 'include', 'alq.h')).read()
    assert re.search(r'int32_t layer;\s*int32_t pad;\s*int64_t idx;\s*double delta;\s*}\s*alq_flip_t;', hdr)
    r = ref64.Ref64.__new__(ref64.Ref64)
    r.L = 5
    r.sizes = np.array([10., 2000., 33., 7., 4098.])
    rs = np.random.RandomState(3)
    z = rs.randn(40, 2) * 4
    z[0] = (-30., 30.)          # p1 > 1 - 1e-6
    z[1] = (30., -30.)          # p1 < 1e-6
    S = rs.randn(40, 5) * 100
    out = r.scores(z, S, diag_load=1e-3)
    p = np.exp(z - z.max(1, keepdims=True))
    p1 = p[:, 1] / p.sum(1)
    g0, g1, A = factored_ref.fisher_from_unit(p1, S, r.sizes, 1e-3)
    g0[1 - p1 < 1e-6] = 0.          # (fisher_from_unit zeroes the skipped class inside A only)
    g1[p1 < 1e-6] = 0.
    np.testing.assert_allclose(out['p1'], p1, rtol=1e-14)
    np.testing.assert_allclose(out['g0'], g0, rtol=1e-13, atol=0)
    np.testing.assert_allclose(out['g1'], g1, rtol=1e-13, atol=0)
    np.testing.assert_allclose(out['A'], A, rtol=1e-12, atol=1e-18)
    assert ref64.Ref64.close(np.array([1.0, 2.0]), np.array([1.0 + 1e-6, 2.0]))
    assert not ref64.Ref64.close(np.array([1.0, 2.0]), np.array([1.001, 2.0]))


def test_bench_clock_sampler_reads_the_hwmon_files_of_the_device(tmp_path, monkeypatch):
    """bench.py's ClockSampler: the device's hwmon directory is found through its PCI address, clock and power are averaged over the
    samples, and a box without the files (or a torch build without the PCI fields) yields None instead of a number."""
    import glob as _glob
    import importlib.util
    import time
    import types
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pci = tmp_path / 'devices' / '0000:f4:00.0'
    hw = pci / 'hwmon' / 'hwmon7'
    hw.mkdir(parents=True)
    (hw / 'freq1_input').write_text('2016000000\n')
    (hw / 'power1_input').write_text('1355000000\n')
    (hw / 'power1_cap').write_text('1400000000\n')
    card = tmp_path / 'card3'
    card.mkdir()
    os.symlink(str(pci), str(card / 'device'))
    real_glob = _glob.glob
    monkeypatch.setattr(_glob, 'glob', lambda p: real_glob(p.replace('/sys/class/drm', str(tmp_path))))
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0xf4, pci_device_id=0)
    torch_like = types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda i: props))
    s = bench.ClockSampler(torch_like, 0)
    assert os.path.realpath(s.dir) == os.path.realpath(str(hw))
    s.start()
    time.sleep(0.12)
    r = s.stop()
    assert r['samples'] >= 2 and r['sclk_mhz_mean'] == 2016.0 and r['power_w_mean'] == 1355.0 and r['power_cap_w'] == 1400.0
    assert r['device'] == '0000:f4:00.0'
    # no such device / no PCI fields: no sampler, no numbers
    other = types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda i: types.SimpleNamespace(pci_bus_id=1, pci_device_id=0)))
    s2 = bench.ClockSampler(other, 0)
    s2.start()
    assert s2.dir is None and s2.stop() is None
    s3 = bench.ClockSampler(types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda i: types.SimpleNamespace())), 0)
    assert s3.dir is None


def test_pass_cut_is_even_equal_and_independent_of_the_pipelines(monkeypatch):
    """DeviceModel.pass_cut (what fisher_device and bench.py's bookkeeping share): an even number of equal passes of at most max_batch
    patches once a call spans several; one pass otherwise; ALQ_NO_PASS_BALANCE restores max_batch-sized passes with a ragged last one;
    the cut never depends on the number of scoring pipelines (the pass-ordered sum of A is then the same bits with one or two)."""
    m = device.DeviceModel.__new__(device.DeviceModel)
    m.max_batch = 2047
    for lanes in (1, 2, 3):
        m.lanes = lanes
        step, starts = m.pass_cut(100000)
        assert (step, len(starts)) == (2000, 50) and starts[-1] + step >= 100000
    assert m.pass_cut(2047) == (2047, [0]) and m.pass_cut(5) == (2047, [0])
    step, starts = m.pass_cut(2048)
    assert (step, starts) == (1024, [0, 1024])
    step, starts = m.pass_cut(3 * 2047)              # three full passes -> four equal ones
    assert len(starts) == 4 and step == -(-3 * 2047 // 4) and all(b - a == step for a, b in zip(starts, starts[1:]))
    for n in (4095, 10000, 16384, 99999, 1000000):
        step, starts = m.pass_cut(n)
        assert len(starts) % 2 == 0 and step <= m.max_batch and (len(starts) - 1) * step < n <= len(starts) * step
    monkeypatch.setenv('ALQ_NO_PASS_BALANCE', '1')
    step, starts = m.pass_cut(100000)
    assert step == 2047 and len(starts) == 49
    monkeypatch.delenv('ALQ_NO_PASS_BALANCE')
    monkeypatch.setenv('ALQ_PASS_MULT', '6')
    assert len(m.pass_cut(100000)[1]) == 54
