"""Host-side rows added in round 2 against goldens produced by the reference's own code
(tests/golden/make_golden_r2.py): volume statistics + grid indices (8f-4), fine-tune batching and loop state (8f-2),
feature refinement and the SDP's problem statement for lambda = 0 and lambda > 0 (8f-1).  CPU only."""
import os

import numpy as np
import pytest

import nnal_amd  # noqa: F401
from nnal_amd import NN, NNAL_tools, PW_AL, PW_NNAL, nrrd_io


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'r2_host.npz'))


def _paths(g, tmp_path=None):
    """The two subjects of the golden, as in-memory arrays or as NRRD files on disk."""
    subs = []
    for s_ in range(2):
        names = ['sub%d_mod0.nrrd' % s_, 'sub%d_mod1.nrrd' % s_, 'sub%d_mask.nrrd' % s_]
        if tmp_path is None:
            subs.append([g['vol_' + n] for n in names])
        else:
            sub = []
            for k, n in enumerate(names):
                p = str(tmp_path / n)
                nrrd_io.write(p, g['vol_' + n], 'gzip' if k == 1 else 'raw')
                sub.append(p)
            subs.append(sub)
    return subs


@pytest.mark.parametrize('on_disk', [False, True])
def test_get_stats_and_grid_indices_vs_reference(g, tmp_path, on_disk):
    paths = _paths(g, tmp_path if on_disk else None)
    np.testing.assert_array_equal(PW_AL.get_stats(paths), g['stats'])
    for sp in (2, 3):
        inds, labels = PW_AL.gen_multimg_inds(paths, sp)
        for i in range(2):
            np.testing.assert_array_equal(np.array(inds[i]), g['grid%d_inds_%d' % (sp, i)])
            np.testing.assert_array_equal(np.array(labels[i]), g['grid%d_labels_%d' % (sp, i)])
            assert not np.isnan(labels[i]).any()
    np.testing.assert_array_equal(PW_AL.get_stats([[paths[1][0], paths[1][2]]]), g['stats_m1'])
    assert bool(g['stats_m3_raises'])
    with pytest.raises(IndexError):        # the reference's [i, j*m] indexing runs out of columns for m = 3: mirrored
        PW_AL.get_stats([[paths[0][0], paths[0][1], paths[0][0], paths[0][2]]])


def test_nrrd_reader_contract(tmp_path):
    rs = np.random.RandomState(1)
    for dt in (np.float32, np.float64, np.int16, np.uint8):
        a = (rs.randn(4, 5, 3) * 50).astype(dt)
        for enc in ('raw', 'gzip'):
            p = str(tmp_path / ('v_%s_%s.nrrd' % (np.dtype(dt).name, enc)))
            nrrd_io.write(p, a, enc)
            b, h = nrrd_io.read(p)
            assert b.dtype == a.dtype and b.shape == (4, 5, 3) and list(h['sizes']) == [4, 5, 3]
            np.testing.assert_array_equal(a, b)
    # first axis fastest in the file (pynrrd's default index order)
    p = str(tmp_path / 'order.nrrd')
    with open(p, 'wb') as f:
        f.write(b'NRRD0004\n# comment\ntype: short\ndimension: 2\nsizes: 3 2\nendian: big\nencoding: raw\nkey:=value\n\n')
        f.write(np.arange(6, dtype='>i2').tobytes())
    b, h = nrrd_io.read(p)
    np.testing.assert_array_equal(b, np.array([[0, 3], [1, 4], [2, 5]]))
    assert h['key'] == 'value'
    with open(p, 'wb') as f:
        f.write(b'NRRD0004\ntype: float\ndimension: 1\nsizes: 2\nencoding: ascii\n\n1 2\n')
    with pytest.raises(NotImplementedError):
        nrrd_io.read(p)
    with open(p, 'wb') as f:
        f.write(b'not a nrrd\n')
    with pytest.raises(ValueError):
        nrrd_io.read(p)


def test_gen_batch_inds_vs_reference(g):
    np.random.seed(31)
    for k in range(3):
        n, b = g['batches_%d_n' % k]
        bt = NN.gen_batch_inds(int(n), int(b))
        np.testing.assert_array_equal([len(x) for x in bt], g['batches_%d_lens' % k])
        np.testing.assert_array_equal([i for x in bt for i in x], g['batches_%d_flat' % k])


def test_refine_feature_matrix_vs_reference(g, capsys):
    out = PW_NNAL.refine_feature_matrix(g['refine_F'].copy(), 16)
    np.testing.assert_array_equal(out, g['refine_out'])
    assert np.linalg.matrix_rank(out) == out.shape[0] and np.linalg.cond(out) <= 1e6


def test_loop_state_files(tmp_path):
    st = PW_AL.LoopState(str(tmp_path / 'fi'))
    assert st.iters_done() == 0
    st.save_round(0, np.array([[17, 0], [4, 1]]), 0.25)
    st.save_round(1, np.array([[9, 1]]), 0.5)
    assert st.iters_done() == 2
    np.testing.assert_array_equal(st.load_queries(), [[17, 0], [4, 1], [9, 1]])
    assert open(os.path.join(st.root, 'queries', '0')).read() == '17 0\n4 1\n'
    assert abs(float(np.loadtxt(os.path.join(st.root, 'AL_running_times', 'dt_1'))) - 0.5) < 1e-12
    st2 = PW_AL.LoopState(str(tmp_path / 'fi'))                  # resume = count the files (PW_AL.py:724-735)
    assert st2.iters_done() == 2 and st2.weights_path(2).endswith('curr_weights_2.npz')


# ------------------------------------------------------------------------------------------ the SDP's statement
def _vec(Z):
    return np.ravel(Z, order='F')          # cvxopt's column-major 'vec' of a dense block


@pytest.mark.parametrize('tag', ['l0', 'l1'])
def test_sdp_solution_certified_against_the_reference_statement(golden_dir, tag):
    """c, G_k, h_k, A, b below were assembled by the REFERENCE's SDP_query_distribution / inequality_cvx_matrix /
    append_zero (cvxopt's matrix / solvers.sdp bound to recording stand-ins, tests/golden/make_golden_r2.py).
    cvxopt's standard form:  min c^T x  s.t.  G_k x + s_k = h_k, s_k >= 0 (PSD),  A x = b;
    dual:  max -sum_k tr(h_k z_k) - b^T y  s.t.  sum_k G_k^T vec(z_k) + A^T y + c = 0,  z_k >= 0.
    This build's (q, t) must be primal feasible for those matrices, and the dual point built from it - z_j = v_j v_j^T
    with v_j = (M^-1 e_j, -1) for the L Schur blocks, z = diag(s) for the positivity block, y from the solver - dual
    feasible, with zero duality gap: that certifies optimality for the reference's own statement of the problem."""
    s = np.load(os.path.join(golden_dir, 'r2_sdp.npz'))
    A, X = s['A'], s['X_pool']
    lam = float(s[tag + '_lambda'])
    n, L = A.shape[0], A.shape[1]
    soln = NNAL_tools.SDP_query_distribution(A, lam, X if lam > 0 else [], 3, tol=1e-10)
    assert soln['status'].startswith('optimal'), soln['status']
    x = np.asarray(soln['x'])
    q, t = x[:n], x[n:]
    c, Aeq, b = s[tag + '_c'].ravel(), s[tag + '_A'], s[tag + '_b'].ravel()
    nG = int(s[tag + '_nG'])
    assert nG == L + 1 and c.shape == (n + L,)
    # ---- primal feasibility in the reference's matrices
    np.testing.assert_allclose(Aeq @ x, b, atol=1e-9)
    for k in range(nG):
        G, h = s[tag + '_G%d' % k], s[tag + '_h%d' % k]
        S = h - (G @ x).reshape(h.shape, order='F')
        np.testing.assert_allclose(S, S.T, atol=1e-12)
        assert np.linalg.eigvalsh(S).min() >= -1e-9, (k, np.linalg.eigvalsh(S).min())
    # ---- dual certificate
    M = np.tensordot(q, A, axes=(0, 0))
    Mi = np.linalg.inv(M)
    zs = []
    for j in range(L):
        v = np.concatenate((Mi[:, j], [-1.0]))
        zs.append(np.outer(v, v))
    d = np.tensordot(A, Mi @ Mi, axes=([1, 2], [0, 1]))
    y = np.asarray(soln['y'], dtype=np.float64)
    if lam > 0:
        sl = -d + c[:n] + Aeq[:, :n].T @ y
    else:
        sl = y[0] - d
    assert sl.min() >= -1e-6 * d.max()
    zs.append(np.diag(np.maximum(sl, 0.)))
    resid = c + Aeq.T @ y if lam > 0 else c + Aeq.T.ravel() * y[0]
    for k in range(nG):
        resid = resid + s[tag + '_G%d' % k].T @ _vec(zs[k])
    assert np.abs(resid).max() <= 1e-6 * max(1., d.max()), np.abs(resid).max()
    primal = float(c @ x)
    dual = -sum(np.sum(s[tag + '_h%d' % k] * zs[k]) for k in range(nG)) - float(b @ y)
    assert abs(primal - soln['primal objective']) <= 1e-9 * max(1., abs(primal))
    assert abs(primal - dual) <= 1e-6 * max(1., abs(primal)), (primal, dual)
    assert np.all(q >= 0) and abs(q.sum() - 1) < 1e-12
    np.testing.assert_array_equal(NNAL_tools.append_zero(A[0]), s['append_zero'])


def test_sdp_lambda_moves_mass_to_long_features_and_needs_centred_rows():
    rs = np.random.RandomState(3)
    n, L = 40, 4
    g_ = rs.randn(n, L) * 0.2
    A = np.stack([np.outer(v, v) + 1e-3 * np.eye(L) for v in g_])
    X = rs.randn(3, n)
    X -= X.mean(axis=1, keepdims=True)
    w = np.sum(X ** 2, axis=0)
    s0 = NNAL_tools.SDP_query_distribution(A, 0., None, 5)
    s1 = NNAL_tools.SDP_query_distribution(A, 5.0, X, 5)
    q0, q1 = s0['x'][:n], s1['x'][:n]
    assert np.abs(X @ q1).max() < 1e-9 and abs(q1.sum() - 1) < 1e-12 and q1.min() >= 0
    assert w @ q1 > w @ q0 - 1e-12 or np.abs(X @ q0).max() > 1e-6          # the reward term acts (q0 need not satisfy X q = 0)
    assert s1['gap'] < 1e-5
    with pytest.raises(ValueError):
        NNAL_tools.SDP_query_distribution(A, 1.0, X + 1.0, 5)


def test_parameter_files_round_trip_numpy_values_and_refuse_everything_else(tmp_path):
    """`parameters.txt` (PW_AL.py:91-113): what `save_parameters` writes, `load_parameters` reads back - numpy scalars and arrays
    (the reference stores `pars['stats']`; np.float64 learning rates are common) as plain values, tuples as tuples, OrderedDict
    as dict.  A file the reference's own `yaml.dump` wrote with numpy tags loads too (decoded from its bytes, nothing called).
    A value that cannot be stored fails at WRITE time (on the writing rank, before the barrier its peers wait in) and leaves no
    file behind; a file with any other python tag is refused with a ValueError that names the file."""
    import collections
    import yaml
    pars = {'patch_shape': (5, 5, 3), 'learning_rate': np.float64(1e-3), 'k': np.int64(4), 'flag': np.bool_(True),
            'stats': np.array([[0.25, 1.5], [0.3, 1.25]]), 'nested': collections.OrderedDict([('b', 2), ('a', (1, np.float32(0.5)))]),
            'grad_layers': [], 'model_name': 'PW'}
    root = str(tmp_path / 'e1')
    e = PW_AL.Experiment(root, pars)
    e2 = PW_AL.Experiment(root)
    e2.load_parameters()
    got = e2.pars
    assert got['patch_shape'] == (5, 5, 3) and isinstance(got['patch_shape'], tuple)
    assert got['learning_rate'] == 1e-3 and type(got['learning_rate']) is float
    assert got['k'] == 4 and type(got['k']) is int and got['flag'] is True
    assert got['stats'] == [[0.25, 1.5], [0.3, 1.25]]
    assert got['nested'] == {'b': 2, 'a': (1, 0.5)} and got['model_name'] == 'PW' and got['grad_layers'] == []
    text = open(os.path.join(root, 'parameters.txt')).read()
    assert 'numpy' not in text and 'python/object' not in text
    # a file as the reference's yaml.dump writes it (numpy tags)
    os.makedirs(str(tmp_path / 'e2'))
    with open(str(tmp_path / 'e2' / 'parameters.txt'), 'w') as f:
        yaml.dump(pars, f)
    assert 'python/object/apply:numpy' in open(str(tmp_path / 'e2' / 'parameters.txt')).read()
    e3 = PW_AL.Experiment(str(tmp_path / 'e2'))
    e3.load_parameters()
    assert e3.pars['learning_rate'] == 1e-3 and e3.pars['k'] == 4 and e3.pars['patch_shape'] == (5, 5, 3)
    np.testing.assert_array_equal(np.asarray(e3.pars['stats']), pars['stats'])
    assert e3.pars['nested'] == {'b': 2, 'a': (1, 0.5)}
    # a value that cannot be stored: TypeError at write time, no file
    with pytest.raises(TypeError, match='bad'):
        PW_AL.Experiment(str(tmp_path / 'e3'), {'bad': {1, 2}})
    assert not os.path.exists(str(tmp_path / 'e3' / 'parameters.txt'))
    # a file that asks for anything else is refused, not executed
    os.makedirs(str(tmp_path / 'e4'))
    with open(str(tmp_path / 'e4' / 'parameters.txt'), 'w') as f:
        f.write("a: !!python/object/apply:os.getcwd []\n")
    with pytest.raises(ValueError, match='parameters.txt'):
        PW_AL.Experiment(str(tmp_path / 'e4')).load_parameters()
