"""Representativeness strategies of query_multimg on the device against a NumPy restatement of the
reference's blocks run on the device's own feature vectors (GPU box)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import netspec  # noqa: E402
from tests.test_oracle_golden import Expr  # noqa: E402


@pytest.fixture(scope='module')
def sess():
    import nnal_amd  # noqa: F401
    from nnal_amd import device
    return device.default_session()


def _setup(sess):
    from nnal_amd import NN
    rs = np.random.RandomState(18)
    patch_shape = (5, 5, 3)
    vols = []
    for s_ in range(2):
        shp = (9 + s_, 10, 8)
        mods = [np.pad(rs.randn(*shp), [(2, 2), (2, 2), (1, 1)], 'constant') for _ in range(2)]
        vols.append(mods + [rs.randint(0, 2, size=shp)])
    pools = [np.sort(rs.permutation(9 * 10 * 8)[:170]), np.sort(rs.permutation(10 * 10 * 8)[:110])]
    labeled = [np.sort(rs.permutation(9 * 10 * 8)[:30]), np.sort(rs.permutation(10 * 10 * 8)[:45])]
    stats = np.array([[0., 1., 0.1, 0.9], [0.05, 1.1, 0., 1.]])
    expr = Expr({'patch_shape': patch_shape, 'ntb': 64, 'k': 9, 'B': 40}, train_stats=stats)
    expr.labeled_stats = stats
    expr.train_paths = expr.labeled_paths = [['a'], ['b']]
    ld = netspec.net_b_small()
    in_shape = (5, 5, 6)
    model = NN.CNN(in_shape, ld, 'rep', len(ld) - 2, None, sess=sess, max_batch=64)
    model.set_weights(netspec.he_init(ld, in_shape, seed=48, bias_std=0.1))
    return expr, model, vols, pools, labeled


def _feats(expr, model, sess, vols, inds_list, stats_attr='train_stats'):
    from nnal_amd import PW_NN
    st = getattr(expr, stats_attr)
    out = []
    for i, inds in enumerate(inds_list):
        stats = [[st[i, 2 * j], st[i, 2 * j + 1]] for j in range(2)]
        out.append(PW_NN.batch_eval(model, sess, vols[i][:-1], inds, expr.pars['patch_shape'], expr.pars['ntb'], stats, 'feature_layer')[0])
    return out


def test_rep_entropy_vs_numpy_restatement(sess):
    from nnal_amd import PW_NNAL, patch_utils
    expr, model, vols, pools, labeled = _setup(sess)
    got = PW_NNAL.query_multimg(expr, model, sess, vols, pools, labeled, 'rep-entropy')
    # PW_NNAL.py:284-351 on the device's features
    F = _feats(expr, model, sess, vols, pools)
    B, k = expr.pars['B'], expr.pars['k']
    sel_inds, _ = PW_NNAL.bin_uncertainty_filter_multimg(expr, model, sess, vols, pools, B)
    F_unc = np.concatenate([F[i][:, sel_inds[i]] for i in range(2) if len(sel_inds[i]) > 0], axis=1)
    F_rem = np.concatenate([F[i][:, np.setdiff1d(np.arange(len(pools[i])), sel_inds[i])] for i in range(2)], axis=1)
    sims = (F_rem.T @ F_unc) / np.outer(np.sqrt((F_rem ** 2).sum(0)), np.sqrt((F_unc ** 2).sum(0)))
    Q, nQ = [], np.arange(B)
    for i in range(k):
        rep = np.array([np.sum(np.max(sims[:, Q + [nQ[j]]], axis=1)) for j in range(B - i)])
        Q += [nQ[np.argmax(rep)]]
        nQ = np.delete(nQ, np.argmax(rep))
    local = patch_utils.global2local_inds(Q, [len(s_) for s_ in sel_inds])
    want = [np.array(sel_inds[i])[local[i]] for i in range(2)]
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    assert sum(len(a) for a in got) == k
    model.close()


def test_core_set_vs_numpy_restatement(sess):
    from nnal_amd import PW_NNAL, patch_utils
    expr, model, vols, pools, labeled = _setup(sess)
    np.random.seed(5)
    got = PW_NNAL.query_multimg(expr, model, sess, vols, pools, labeled, 'core-set')
    # PW_NNAL.py:353-451: labelled side = the last subject only (loop variable read after its loop)
    F_u = np.concatenate(_feats(expr, model, sess, vols, pools), axis=1)
    F_T = _feats(expr, model, sess, vols, labeled, 'labeled_stats')[1]
    norms_u = np.sqrt((F_u ** 2).sum(0))
    sims = np.max((F_T.T @ F_u) / np.outer(np.sqrt((F_T ** 2).sum(0)), norms_u), axis=0)
    Q = []
    for _ in range(expr.pars['k']):
        q = int(np.argmin(sims))
        Q.append(q)
        sims = np.maximum(sims, (F_u[:, q] @ F_u) / (norms_u * norms_u[q]))
        sims[q] = np.inf
    want = patch_utils.global2local_inds(Q, [len(p) for p in pools])
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    model.close()


def test_similarity_kernels_vs_numpy(sess):
    """alq_row_norms / alq_cosine_sims / alq_colsum_max / alq_take_colmax / alq_fold_rowmax on ragged sizes."""
    import ctypes as C
    from nnal_amd import PW_NNAL
    from nnal_amd._lib import check
    torch = sess.torch
    rs = np.random.RandomState(19)
    A = rs.randn(333, 70).astype(np.float32)
    Bm = rs.randn(45, 70).astype(np.float32)
    dA, dB = sess.to_device(A, torch.float32), sess.to_device(Bm, torch.float32)
    na, nb = PW_NNAL._row_norms(sess, dA), PW_NNAL._row_norms(sess, dB)
    np.testing.assert_allclose(na.cpu().numpy(), np.sqrt((A.astype(np.float64) ** 2).sum(1)), rtol=1e-14)
    S = PW_NNAL._cosine_sims(sess, dA, na, dB, nb)
    ref = (A.astype(np.float64) @ Bm.astype(np.float64).T) / np.outer(na.cpu().numpy(), nb.cpu().numpy())
    np.testing.assert_allclose(S.cpu().numpy(), ref, rtol=1e-12, atol=1e-15)
    cmax = sess.to_device(rs.randn(333) * 0.1, torch.float64)
    work = sess.empty((sess.lib.alq_colsum_work_bytes(333, 45),), torch.uint8)
    out = sess.empty((45,), torch.float64)
    check(sess.lib.alq_colsum_max(sess.ctx, C.c_void_p(S.data_ptr()), 333, 45, C.c_void_p(cmax.data_ptr()), None, C.c_void_p(out.data_ptr()),
                                  C.c_void_p(work.data_ptr())))
    np.testing.assert_allclose(out.cpu().numpy(), np.maximum(cmax.cpu().numpy()[:, None], ref).sum(0), rtol=1e-12)
    check(sess.lib.alq_take_colmax(sess.ctx, C.c_void_p(S.data_ptr()), 333, 45, 7, 0, C.c_void_p(cmax.data_ptr())))
    v = sess.to_device(np.full(45, -np.inf), torch.float64)
    check(sess.lib.alq_fold_rowmax(sess.ctx, C.c_void_p(S.data_ptr()), 333, 45, C.c_void_p(v.data_ptr())))
    np.testing.assert_allclose(v.cpu().numpy(), ref.max(0), rtol=1e-12)
