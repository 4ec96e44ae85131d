"""Round-3 goldens (tests/golden/r3_*.npz, produced by tests/golden/make_golden_r3.py from the reference's own code):

  * NET-B (NN.create_PW1's layer dict) at the benchmarked input [N, 32, 32, 32]: oracle bit for bit (CPU), device (GPU);
  * query_multimg 'rep-entropy' / 'core-set': the device strategies pick the reference's queries (GPU);
  * PW_AL.finetune / finetune_multimg: the feeds of every train_step - batch order, patches, one-hot labels, keep_prob -
    bit for bit, with the oracle's gather on the CPU and the device gather on the GPU;
  * NNAL.CNN_query(..., 'fi') (image-level, multi-class): class selection + A matrices + refined features + queries,
    host logic against an oracle-backed model on the CPU, the device path on the GPU.
"""
import os

import numpy as np
import pytest

from oracle import alpath, netspec
from oracle.model import OracleModel, OracleSession
from tests.test_oracle_golden import Expr


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ------------------------------------------------------------------------------------------------ NET-B at 32 channels
def _netb32(g):
    ld = netspec.net_b()
    in_shape = tuple(int(v) for v in g['in_shape'])
    pars = netspec.he_init(ld, in_shape, seed=int(g['wseed']), bias_std=float(g['bias_std']))
    x = np.random.RandomState(int(g['xseed'])).randn(int(g['n']), *in_shape).astype(np.float32)
    np.testing.assert_array_equal(x[:, :2, :2, :4], g['x_head'])
    return ld, in_shape, pars, x


def test_oracle_netb_32ch_vs_golden(golden_dir):
    g = _load(golden_dir, 'r3_fisher_netb_32ch.npz')
    ld, in_shape, pars, x = _netb32(g)
    om = OracleModel(ld, in_shape, pars)
    n = 3                                               # the 42 M-parameter net: three patches keep the CPU suite short
    p1 = om.forward(x[:n])['posteriors'][1].astype(np.float64)
    np.testing.assert_array_equal(p1, g['p1'][:n])
    A = alpath.gen_A_matrices(Expr({'patch_shape': in_shape}), om, OracleSession(om), x[:n], p1, float(g['diag_load']))
    np.testing.assert_array_equal(np.stack(A), g['A'][:n])


@pytest.mark.gpu
@pytest.mark.parametrize('max_batch', [4, 3, 16])
def test_device_netb_32ch_vs_golden(golden_dir, max_batch):
    """The bench's `netb` side figure is quoted on this shape: igemm3's 5x5 conv with 32 input channels, fcgemm with
    K = 6144.  max_batch 3 walks the 8 patches in passes of 3, 3, 2."""
    from nnal_amd import device
    from tests.test_gpu_parity import assert_scores_close, P_ATOL, SCORE_ATOL, G_RTOL, A_RTOL
    g = _load(golden_dir, 'r3_fisher_netb_32ch.npz')
    ld, in_shape, pars, x = _netb32(g)
    sess = device.default_session()
    model = device.DeviceModel(sess, ld, in_shape, (), max_batch=max_batch)
    model.set_weights(pars)
    res = model.fisher(x, g['p1'], float(g['diag_load']))
    np.testing.assert_allclose(res['p1'], g['p1'], rtol=0, atol=P_ATOL)
    assert_scores_close(res['g0'], g['g0'], SCORE_ATOL, G_RTOL, 1e-5)
    assert_scores_close(res['g1'], g['g1'], SCORE_ATOL, G_RTOL, 1e-5)
    assert_scores_close(res['A'], g['A'], SCORE_ATOL * 0.1, A_RTOL, 1e-6)
    assert_scores_close(res['Asum'], g['A'].sum(0), SCORE_ATOL, 5 * A_RTOL, 1e-6)
    model.close()


# ------------------------------------------------------------------------------------------------ fine-tune feeds
class _Recorder(object):
    def __init__(self):
        self.feeds = []

    def run(self, fetch, feed_dict=None):
        assert fetch == 'train_step'
        self.feeds.append(feed_dict)


class _TrainModel(object):
    train_step, x, y_, keep_prob = 'train_step', 'x', 'y_', 'keep_prob'
    dropout_rate = 0.5


def _check_feeds(g, tag, feeds):
    lens = [f['x'].shape[0] for f in feeds]
    np.testing.assert_array_equal(lens, g[tag + '_lens'])
    np.testing.assert_array_equal(np.concatenate([f['x'] for f in feeds]), g[tag + '_x'])
    np.testing.assert_array_equal(np.concatenate([f['y_'] for f in feeds], axis=1), g[tag + '_y'])
    np.testing.assert_array_equal([f['keep_prob'] for f in feeds], g[tag + '_kp'])


def _run_finetunes(g, PW_AL):
    pshape = tuple(int(v) for v in g['pshape'])
    subs = [[g['sub%d_%d' % (s_, j)] for j in range(3)] for s_ in range(2)]
    model = _TrainModel()
    b, ep = [int(v) for v in g['ft_b_epochs']]
    expr = Expr({'patch_shape': pshape, 'b': b, 'epochs': ep, 'stats': g['ft_stats'].tolist()})
    rec = _Recorder()
    np.random.seed(int(g['ft_seed']))
    PW_AL.finetune(model, rec, expr, subs[0][:2], subs[0][2], g['ft_train_inds'])
    _check_feeds(g, 'ft', rec.feeds)
    b, ep = [int(v) for v in g['fm_b_epochs']]
    expr = Expr({'patch_shape': pshape, 'b': b, 'epochs': ep}, train_stats=g['fm_tstats'])
    rec = _Recorder()
    np.random.seed(int(g['fm_seed']))
    PW_AL.finetune_multimg(expr, model, rec, subs, [list(g['fm_train_inds_%d' % s_]) for s_ in range(2)])
    _check_feeds(g, 'fm', rec.feeds)


def test_finetune_feeds_host_logic(golden_dir, monkeypatch):
    """PW_AL.finetune / finetune_multimg (PW_AL.py:1030-1147) with the oracle's gather standing in for the device gather:
    batch order (NN.gen_batch_inds on the global stream), global -> local index split, channel-index vs slab
    normalisation, one-hot labels and keep_prob = model.dropout_rate are the reference's, bit for bit."""
    import nnal_amd  # noqa: F401
    from nnal_amd import PW_AL, patch_utils
    monkeypatch.setattr(patch_utils, 'get_patches', alpath.get_patches)
    monkeypatch.setattr(patch_utils, 'get_patches_multimg', alpath.get_patches_multimg)
    _run_finetunes(_load(golden_dir, 'r3_finetune.npz'), PW_AL)


@pytest.mark.gpu
def test_finetune_feeds_device_gather(golden_dir):
    import nnal_amd  # noqa: F401
    from nnal_amd import PW_AL
    _run_finetunes(_load(golden_dir, 'r3_finetune.npz'), PW_AL)


# ------------------------------------------------------------------------------------------------ representativeness
@pytest.mark.gpu
def test_rep_entropy_and_core_set_vs_reference_run(golden_dir):
    """query_multimg 'rep-entropy' / 'core-set' (PW_NNAL.py:284-451) executed by the reference's own code against the
    oracle (golden) vs the device strategies: same queries."""
    import nnal_amd  # noqa: F401
    from nnal_amd import NN, PW_NNAL, device
    g = _load(golden_dir, 'r3_strategies.npz')
    sess = device.default_session()
    pshape = tuple(int(v) for v in g['pshape'])
    subs = [[g['sub%d_%d' % (s_, j)] for j in range(3)] for s_ in range(2)]
    pools = [g['pool_%d' % s_] for s_ in range(2)]
    labeled = [g['labeled_%d' % s_] for s_ in range(2)]
    ld = netspec.net_a()
    in_shape = (pshape[0], pshape[1], 2 * pshape[2])
    model = NN.CNN(in_shape, ld, 'strategies', len(ld) - 2, None, sess=sess, max_batch=64)
    model.set_weights(netspec.he_init(ld, in_shape, seed=int(g['wseed']), bias_std=0.05))
    for tag in ('re_a', 're_b'):
        B, k = [int(v) for v in g[tag + '_Bk']]
        expr = Expr({'patch_shape': pshape, 'ntb': 37, 'k': k, 'B': B}, train_stats=g['tstats'])
        Q = PW_NNAL.query_multimg(expr, model, sess, subs, pools, labeled, 'rep-entropy')
        for s_ in range(2):
            np.testing.assert_array_equal(Q[s_], g['%s_Q_%d' % (tag, s_)], err_msg=tag)
    for tag in ('cs_a', 'cs_b'):
        k, seed = [int(v) for v in g[tag + '_k_seed']]
        expr = Expr({'patch_shape': pshape, 'ntb': 37, 'k': k, 'B': 24}, train_stats=g['tstats'])
        expr.labeled_stats = g['tstats']
        expr.train_paths = expr.labeled_paths = ['same']
        np.random.seed(seed)
        Q = PW_NNAL.query_multimg(expr, model, sess, subs, pools, labeled, 'core-set')
        for s_ in range(2):
            np.testing.assert_array_equal(Q[s_], g['%s_Q_%d' % (tag, s_)], err_msg=tag)
    model.close()


# ------------------------------------------------------------------------------------------------ image-level fi
class _OracleImgModel(object):
    """CPU stand-in with the attributes NNAL.CNN_query reads from a device NN.CNN (TEST INFRASTRUCTURE)."""

    def __init__(self, om):
        self.om, self.osess = om, OracleSession(om)
        self.x, self.keep_prob, self.posteriors, self.feature_layer = om.x, om.keep_prob, om.posteriors, om.feature_layer
        self.nclass = om.nclass
        self.feature_dim = om.feature_layer.shape[0].value

    def extract_features(self, inds, expr, session):
        from nnal_amd import NN
        return NN.CNN.extract_features(self, inds, expr, session)

    def fisher_classes(self, x, W, diag):
        n, c = W.shape
        out = []
        for i in range(n):
            Ai = 0.
            for j in range(c):
                if W[i, j] != 0:
                    gj = alpath.shrink_gradient(self.om.grad_log_post(j, x[i:i + 1]))
                    Ai = Ai + np.outer(gj, gj) * W[i, j]
            out.append(Ai + np.eye(self.om.nlayers_par) * diag[i])
        return np.stack(out)


def _imgfi_case(g, tag, tmp_path, make_model, a_rtol, a_atol):
    import nnal_amd  # noqa: F401
    from nnal_amd import NN, NNAL, NNAL_tools
    c, wseed, seed, k, B = [int(v) for v in g[tag + '_meta']]
    imgs = g['imgs']
    pfile = tmp_path / 'paths.txt'
    with open(pfile, 'w') as f:
        for i in range(len(imgs)):
            np.save(tmp_path / ('img_%d.npy' % i), imgs[i])
            f.write(str(tmp_path / ('img_%d.npy' % i)) + '\n')
    hw = imgs.shape[1]
    ld = netspec.net_a(nclass=c)
    in_shape = (hw, hw, 3)
    pars = netspec.he_init(ld, in_shape, seed=wseed, bias_std=0.05)
    last = list(pars.keys())[-1]
    pars[last][0] = (pars[last][0] * float(g[tag + '_logit_scale'])).astype(np.float32)
    model, session = make_model(ld, in_shape, pars)
    expr = Expr({'k': k, 'B': B, 'lambda_': 0.5, 'batch_size': 8, 'target_shape': (hw, hw), 'mean': 100.})
    expr.imgs_path_file = str(pfile)
    rec = {}

    def sdp_recorder(A, lambda_, X_pool, k_):
        rec.update(A=np.stack(A), F=np.array(X_pool), lambda_=lambda_)
        return {'status': 'recorded', 'x': np.ones(len(A) + A[0].shape[0]) / len(A)}
    orig = NNAL_tools.SDP_query_distribution
    NNAL_tools.SDP_query_distribution = sdp_recorder
    try:
        np.random.seed(seed)
        Q = NNAL.CNN_query(model, expr, g[tag + '_pool_inds'], 'fi', session, col=True)
    finally:
        NNAL_tools.SDP_query_distribution = orig
    np.testing.assert_array_equal(Q, g[tag + '_Q'])
    assert rec['lambda_'] == 0.5
    np.testing.assert_allclose(rec['A'], g[tag + '_A'], rtol=a_rtol, atol=a_atol)
    np.testing.assert_allclose(rec['F'], g[tag + '_F'], rtol=0, atol=2e-5 * np.abs(g[tag + '_F']).max())
    return model


@pytest.mark.parametrize('tag', ['c3', 'c12'])
def test_image_level_fi_host_logic(golden_dir, tmp_path, tag):
    """NNAL.CNN_query(..., 'fi') (NNAL.py:315-464) with an oracle-backed model: uncertainty filter, the per-sample class
    selection (c3: classes under 1e-6 dropped; c12: the ten-largest branch), A_i, refined + centred features and the
    seeded draws equal the reference's run (the A matrices to fp64 rounding: another summation order of the classes)."""
    def make(ld, in_shape, pars):
        om = OracleModel(ld, in_shape, pars, feature_layer=len(ld) - 2)
        m = _OracleImgModel(om)
        return m, m.osess
    _imgfi_case(_load(golden_dir, 'r3_imgfi.npz'), tag, tmp_path, make, 1e-12, 1e-18)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['c3', 'c12'])
def test_image_level_fi_device(golden_dir, tmp_path, tag):
    """The same query through the device: per-class gradients (alq_param_grads), alq_shrink_sum, alq_fisher_classes."""
    from nnal_amd import NN, device
    sess = device.default_session()

    def make(ld, in_shape, pars):
        m = NN.CNN(in_shape, ld, 'imgfi', len(ld) - 2, None, sess=sess, max_batch=5)      # 14 candidates: passes of 5, 5, 4
        m.set_weights(pars)
        return m, sess
    g = _load(golden_dir, 'r3_imgfi.npz')
    # A_i = sum_j g g^T / p_j: entries reach 1e3 for classes with p ~ 1e-6, so the bar is relative to the matrix' scale
    model = _imgfi_case(g, tag, tmp_path, make, 2e-3, 2e-5 * np.abs(g[tag + '_A']).max())
    model.close()


@pytest.mark.gpu
def test_shrink_sum_and_fisher_classes_kernels():
    import ctypes as C
    import nnal_amd  # noqa: F401
    from nnal_amd import device
    from nnal_amd._lib import check
    sess = device.default_session()
    torch = sess.torch
    rs = np.random.RandomState(77)
    sizes = [5 * 5 * 3 * 16 + 16, 3 * 3 * 16 * 32 + 32, 7, 1]
    P, N = sum(sizes), 5
    G = (rs.randn(N, P) * 1e-2).astype(np.float32)
    out = sess.empty((N, len(sizes)), torch.float64)
    Gd = sess.to_device(G, torch.float32)
    check(sess.lib.alq_shrink_sum(sess.ctx, C.c_void_p(Gd.data_ptr()), N, P, (C.c_int64 * 4)(*sizes), 4, C.c_void_p(out.data_ptr())))
    off = np.cumsum([0] + sizes)
    ref = np.stack([[G[n, off[t]:off[t + 1]].astype(np.float64).sum() / sizes[t] for t in range(4)] for n in range(N)])
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-13, atol=1e-300)
    c, L = 6, 4
    g = rs.randn(N, c, L)
    W = rs.rand(N, c) * (rs.rand(N, c) > .3)
    diag = rs.rand(N) * 1e-4
    A = sess.empty((N, L, L), torch.float64)
    gd, Wd, dd = (sess.to_device(v, torch.float64) for v in (g, W, diag))
    check(sess.lib.alq_fisher_classes(sess.ctx, C.c_void_p(gd.data_ptr()), C.c_void_p(Wd.data_ptr()), C.c_void_p(dd.data_ptr()), N, c, L,
                                      C.c_void_p(A.data_ptr())))
    refA = np.einsum('nc,ncr,ncs->nrs', W, g, g) + diag[:, None, None] * np.eye(L)
    np.testing.assert_allclose(A.cpu().numpy(), refA, rtol=1e-13, atol=1e-16)
