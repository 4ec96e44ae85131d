"""Parameter gradients, dropout passes and the training step of the device path against the oracle (GPU box).

Rows (a6) get_gradients, (f2) fine-tune / train_step, (f3) MC-dropout strategies and diagonal_Fisher of SURVEY.md 8."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import alpath, netspec  # noqa: E402
from oracle.model import OracleModel, OracleSession  # noqa: E402
from oracle.train import OracleOptimizer  # noqa: E402
from tests.test_oracle_golden import Expr  # noqa: E402


def torch_no_grad():
    import torch
    return torch.no_grad()


def as_tensor(a):
    import torch
    return torch.as_tensor(np.asarray(a))


@pytest.fixture(scope='module')
def sess():
    import nnal_amd  # noqa: F401
    from nnal_amd import device
    return device.default_session()


def _nets():
    ld_c, sk_c = netspec.net_c()
    ld_c2, sk_c2 = netspec.net_c_2d()
    return [('neta', netspec.net_a(), (20, 20, 1), ()),
            ('netb_small', netspec.net_b_small(), (25, 25, 2), ()),
            ('netc2d', ld_c2, (16, 16, 2), sk_c2),
            ('netc', ld_c, (8, 8, 8, 1), sk_c),
            ('netc_12', ld_c, (12, 8, 16, 1), sk_c)]


def _mk(sess, ld, in_shape, sk, seed, max_batch=16, dropout=None):
    from nnal_amd import device
    pars = netspec.he_init(ld, in_shape, seed=seed, skips=sk, bias_std=0.05)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=max_batch, dropout=dropout)
    m.set_weights(pars)
    return m, pars


def _close(dev, ref, rtol=2e-4, name=''):
    """Gradient arrays agree to fp32 rounding relative to the array's scale (a ReLU input within rounding of zero can
    move single entries by more: bounded by 1e-3 of the scale)."""
    for k, (a, b) in enumerate(zip(dev, ref)):
        assert a.shape == b.shape, (name, k, a.shape, b.shape)
        scale = np.abs(b).max() + 1e-30
        err = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
        assert err <= rtol * scale + 1e-9, '%s: array %d (%s) max err %.3e vs scale %.3e' % (name, k, a.shape, err, scale)


@pytest.mark.parametrize('name,ld,in_shape,sk', _nets())
def test_full_log_posterior_gradients_vs_oracle(sess, name, ld, in_shape, sk):
    """sess.run(model.grad_posts[str(j)], {x: one patch}) = tf.gradients(log posteriors[j, 0], variables)
    (NN.py:639-645): every W and b gradient in TF layout against torch autograd of the oracle graph, both classes,
    plus the batched per-sample form and shrink_gradient 'sum' / 'max' on top of them."""
    from nnal_amd import NNAL_tools
    m, pars = _mk(sess, ld, in_shape, sk, 41)
    om = OracleModel(ld, in_shape, pars, skips=sk)
    osess = OracleSession(om)
    x = np.random.RandomState(3).randn(5, *in_shape).astype(np.float32)
    for j in (0, 1):
        for i in (0, 3):
            dev = sess.run(m.grad_posts[str(j)], feed_dict={m.x: x[[i]], m.keep_prob: 1.})
            ref = osess.run(om.grad_posts[str(j)], feed_dict={om.x: x[[i]], om.keep_prob: 1.})
            assert len(dev) == 2 * m.L == len(ref)
            _close(dev, ref, name='%s class %d patch %d' % (name, j, i))
            for method in ('sum', 'max'):
                # 'max' of the head is ill-posed: its bias gradient is (e_j - p) = (+q, -q), an exact tie in magnitude
                sl = slice(None) if method == 'sum' else slice(0, m.L - 1)
                np.testing.assert_allclose(NNAL_tools.shrink_gradient(dev, method)[sl], NNAL_tools.shrink_gradient(ref, method)[sl],
                                           rtol=2e-3, atol=1e-7)
    # batched: one gradient per sample in one call
    t = sess.to_device(x.reshape(5, -1), sess.torch.float32)
    g, post, _ = m.param_grads_device(t, 5, 0, cls=1, want_post=True)
    g = g.cpu().numpy()
    for i in range(5):
        _close(m.unflatten(g[i]), om.grad_log_post(1, x[[i]]), name='%s batched %d' % (name, i))
    np.testing.assert_allclose(post.cpu().numpy(), om.forward(x)['posteriors'], rtol=0, atol=2e-5)
    m.close()


def test_grad_layers_subset_changes_the_A_matrices(sess):
    """get_gradients(grad_layers) (NN.py:627-633): grad_posts lists only those layers, A_size follows (PW_NNAL.py:751)."""
    from nnal_amd import PW_NNAL
    ld, sk = netspec.net_c()
    in_shape = (8, 8, 8, 1)
    m, pars = _mk(sess, ld, in_shape, sk, 42)
    x = np.random.RandomState(4).randn(6, *in_shape).astype(np.float32)
    p = m.forward(x)['posteriors'][1].astype(np.float64)
    full = np.stack(PW_NNAL.gen_A_matrices(Expr({'patch_shape': in_shape[:3]}), m, sess, x, p, 1e-3))
    names = [m.var_names[2], m.var_names[5], m.var_names[7]]
    m.get_gradients(names)
    assert len(m.grad_posts['1']) == 6
    sub = np.stack(PW_NNAL.gen_A_matrices(Expr({'patch_shape': in_shape[:3]}), m, sess, x, p, 1e-3))
    assert sub.shape == (6, 3, 3)
    idx = [2, 5, 7]
    np.testing.assert_allclose(sub, full[:, idx][:, :, idx], rtol=1e-12, atol=1e-18)
    g = sess.run(m.grad_posts['0'], feed_dict={m.x: x[[0]], m.keep_prob: 1.})
    assert [a.shape for a in g] == [s for t in idx for s in (m.param_shapes[t][1], m.param_shapes[t][2])]
    with pytest.raises(KeyError):
        m.get_gradients(['no_such_layer'])
    m.close()


@pytest.mark.parametrize('opt,lr', [('SGD', 0.003), ('Adam', 0.002)])
@pytest.mark.parametrize('name,ld,in_shape,sk', [_nets()[0], _nets()[3]])
def test_train_step_vs_oracle(sess, name, ld, in_shape, sk, opt, lr):
    """sess.run(model.train_step, {x, y_, keep_prob: 1}) five times (NN.py:583-615) against the oracle's restatement of
    mean softmax cross-entropy + SGD / Adam: losses and every weight after the steps."""
    m, pars = _mk(sess, ld, in_shape, sk, 43)
    m.get_optimizer(lr, [], opt)
    om = OracleModel(ld, in_shape, pars, skips=sk)
    oo = OracleOptimizer(om, lr, (), opt)
    rs = np.random.RandomState(5)
    for step in range(5):
        x = rs.randn(12, *in_shape).astype(np.float32)
        lab = rs.randint(0, 2, size=12)
        y = np.zeros((2, 12))
        y[lab, np.arange(12)] = 1
        l_dev = sess.run(m.train_step, feed_dict={m.x: x, m.y_: y, m.keep_prob: 1.})
        l_ref = oo.step(x, y)
        assert abs(l_dev - l_ref) <= 2e-5 * max(1., abs(l_ref)), (step, l_dev, l_ref)
    for n in m.var_names:
        for a, b in zip(m.var_dict[n], om.params[n]):
            b = b.detach().numpy()
            # Adam divides by sqrt(v): entries whose gradient is ~0 amplify rounding, compare against the step size
            np.testing.assert_allclose(a, b, rtol=0, atol=(5e-4 if opt == 'Adam' else 2e-5) * max(np.abs(b).max(), lr))
    m.close()


@pytest.mark.parametrize('opt,lr', [('SGD', 0.05), ('Adam', 0.002)])
def test_weights_loaded_between_steps_are_what_the_next_step_updates(sess, tmp_path, opt, lr):
    """The TF variables are the single state of the reference: train, load_weights / perform_assign_ops(W0) (PW_AL.py:793-798
    does this between methods and on resume), train again -> the second step starts from W0.  The optimiser's device copy
    of the parameters must follow every set_weights; Adam's slots persist like TF's slot variables."""
    ld = netspec.net_a()
    in_shape = (20, 20, 1)
    m, pars = _mk(sess, ld, in_shape, (), 45, max_batch=8)
    m.get_optimizer(lr, [], opt)
    om = OracleModel(ld, in_shape, pars)
    oo = OracleOptimizer(om, lr, (), opt)
    m.save_weights(str(tmp_path / 'w0.npz'))
    rs = np.random.RandomState(8)

    def batch():
        x = rs.randn(10, *in_shape).astype(np.float32)
        y = np.zeros((2, 10))
        y[rs.randint(0, 2, size=10), np.arange(10)] = 1
        return x, y
    x, y = batch()
    assert abs(m.train_on_batch(x, y) - oo.step(x, y)) < 1e-5
    m.perform_assign_ops(str(tmp_path / 'w0.npz'))                 # back to W0 on the device ...
    with torch_no_grad():
        for n, (w, b) in om.params.items():                         # ... and in the oracle
            w.copy_(as_tensor(pars[n][0]))
            b.copy_(as_tensor(pars[n][1]))
    x, y = batch()
    l_dev, l_ref = m.train_on_batch(x, y), oo.step(x, y)
    assert abs(l_dev - l_ref) < 1e-5, (l_dev, l_ref)
    for n in m.var_names:
        for a, b in zip(m.var_dict[n], om.params[n]):
            b = b.detach().numpy()
            np.testing.assert_allclose(a, b, rtol=0, atol=(5e-4 if opt == 'Adam' else 2e-6) * max(np.abs(b).max(), lr))
    m.close()


def test_train_layers_and_multi_pass_batches(sess):
    """train_layers restricts the update (NN.py:599-615); a batch larger than max_batch is accumulated over passes
    with the loss still the mean over the whole batch."""
    ld = netspec.net_a()
    in_shape = (20, 20, 1)
    m, pars = _mk(sess, ld, in_shape, (), 44, max_batch=8)
    m.get_optimizer(0.1, ['conv2'], 'SGD')
    om = OracleModel(ld, in_shape, pars)
    oo = OracleOptimizer(om, 0.1, ['conv2'], 'SGD')
    rs = np.random.RandomState(6)
    x = rs.randn(21, *in_shape).astype(np.float32)
    y = np.zeros((2, 21))
    y[rs.randint(0, 2, size=21), np.arange(21)] = 1
    before = {n: [w.copy() for w in m.var_dict[n]] for n in m.var_names}
    l_dev = m.train_on_batch(x, y)
    l_ref = oo.step(x, y)
    assert abs(l_dev - l_ref) < 1e-5
    for n in m.var_names:
        for a, b, c in zip(m.var_dict[n], om.params[n], before[n]):
            np.testing.assert_allclose(a, b.detach().numpy(), rtol=0, atol=2e-6)
            if n != 'conv2':
                np.testing.assert_array_equal(a, c)
    m.close()


def test_dropout_forward_and_gradients_vs_oracle(sess):
    """keep_prob < 1: the build's counter-based mask (alq_forward_dropout) against the oracle's restatement of the same
    generator - posteriors, batch-split independence, and the gradient pass through the masks."""
    ld, sk = netspec.net_c()
    in_shape = (8, 8, 8, 1)
    drop_layers = [2, 4, 6, 8]          # conv, conv, conv (after a skip concat), conv
    m, pars = _mk(sess, ld, in_shape, sk, 45, max_batch=4, dropout=[drop_layers, 0.6])
    om = OracleModel(ld, in_shape, pars, skips=sk)
    x = np.random.RandomState(7).randn(10, *in_shape).astype(np.float32)
    t = sess.to_device(x.reshape(10, -1), sess.torch.float32)
    post, _ = m.forward_dropout_device(t, 10, 0.6, seed=1234)
    ref = om.forward(x, drop=dict(layers=drop_layers, keep_prob=0.6, seed=1234))['posteriors']
    np.testing.assert_allclose(post.cpu().numpy(), ref, rtol=0, atol=2e-5)
    assert np.abs(ref - om.forward(x)['posteriors']).max() > 1e-3          # the masks really act
    m2, _ = _mk(sess, ld, in_shape, sk, 45, max_batch=10, dropout=[drop_layers, 0.6])
    post2, _ = m2.forward_dropout_device(t, 10, 0.6, seed=1234)
    assert sess.torch.equal(post, post2)                                    # keyed by sample id, not by batch position
    g, _, _ = m2.param_grads_device(t, 10, 0, cls=0, keep_prob=0.6, seed=99)
    for i in (0, 7):
        refg = om.grad_log_post(0, x[[i]], drop=dict(layers=drop_layers, keep_prob=0.6, seed=99, first_sample=i))
        _close(m2.unflatten(g[i].cpu().numpy()), refg, name='dropout grads %d' % i)
    m.close()
    m2.close()


def test_mc_entropy_and_bald_queries(sess, golden_dir):
    """query_multimg 'MC-entropy' / 'BALD' (PW_NNAL.py:232-282) through batch_eval's x_feed_dict: the same NumPy
    orchestration on the device's dropout posteriors, reproducible under np.random.seed, and equal to the host formulas
    evaluated on those posteriors."""
    from nnal_amd import PW_NNAL, NN
    g = np.load(os.path.join(golden_dir, 'host_layers.npz'))
    rs = np.random.RandomState(8)
    patch_shape = (5, 5, 3)
    vols = []
    for s_ in range(2):
        shp = (9 + s_, 10, 8)
        mods = [np.pad(rs.randn(*shp), [(2, 2), (2, 2), (1, 1)], 'constant') for _ in range(2)]
        vols.append(mods + [rs.randint(0, 2, size=shp)])
    pools = [np.sort(rs.permutation(9 * 10 * 8)[:150]), np.sort(rs.permutation(10 * 10 * 8)[:90])]
    stats = np.array([[0., 1., 0.1, 0.9], [0.05, 1.1, 0., 1.]])
    expr = Expr({'patch_shape': patch_shape, 'ntb': 64, 'k': 12, 'B': 30, 'MC_iters': 4}, train_stats=stats)
    ld = netspec.net_b_small()
    in_shape = (5, 5, 6)
    pars = netspec.he_init(ld, in_shape, seed=46)
    model = NN.CNN(in_shape, ld, 'mc', None, [[1, 3], 0.7], sess=sess, max_batch=64)
    model.set_weights(pars)
    out = {}
    for method in ('MC-entropy', 'BALD'):
        np.random.seed(77)
        q1 = PW_NNAL.query_multimg(expr, model, sess, vols, pools, None, method)
        np.random.seed(77)
        q2 = PW_NNAL.query_multimg(expr, model, sess, vols, pools, None, method)
        for a, b in zip(q1, q2):
            np.testing.assert_array_equal(a, b)
        assert sum(len(a) for a in q1) == 12 and all(len(np.unique(a)) == len(a) for a in q1)
        out[method] = q1
    # host recomputation from the same dropout posteriors
    np.random.seed(77)
    av, av_e = 0, 0
    for i in range(4):
        posts = PW_NNAL.bin_uncertainty_filter_multimg(expr, model, sess, vols, pools, 12, {model.keep_prob: model.dropout_rate})
        av = (posts + i * av) / (i + 1)
        e = -posts * np.log(np.where(posts == 0, 1e-6, posts)) - (1 - posts) * np.log(np.where(1 - posts == 0, 1e-6, 1 - posts))
        av_e = (e + i * av_e) / (i + 1)
    want_mc = np.argsort(np.abs(av - .5), kind='stable')[:12]
    got_mc = np.concatenate([out['MC-entropy'][0], 150 + out['MC-entropy'][1]])
    assert set(got_mc) == set(want_mc)
    ent_av = -av * np.log(av) - (1 - av) * np.log(1 - av)
    want_b = np.argsort(-(ent_av - av_e), kind='stable')[:12]
    got_b = np.concatenate([out['BALD'][0], 150 + out['BALD'][1]])
    assert set(got_b) == set(want_b)
    # MC averaging differs from the deterministic entropy query
    det = PW_NNAL.query_multimg(expr, model, sess, vols, pools, None, 'entropy')
    assert set(np.concatenate([det[0], 150 + det[1]])) != set(got_mc)
    model.close()
    del g


def test_diagonal_fisher_vs_oracle(sess):
    """model_utils.diagonal_Fisher (model_utils.py:294-330): mean squared per-sample gradient of the log-likelihood
    of each sample's label."""
    ld = netspec.net_a()
    in_shape = (20, 20, 1)
    m, pars = _mk(sess, ld, in_shape, (), 47)
    om = OracleModel(ld, in_shape, pars)
    rs = np.random.RandomState(9)
    x = rs.randn(9, *in_shape).astype(np.float32)
    lab = rs.randint(0, 2, size=9)
    from nnal_amd import model_utils
    y = np.zeros((2, 9))
    y[lab, np.arange(9)] = 1
    dev = model_utils.diagonal_Fisher(m, sess, (x, y))
    ref = [np.zeros(a.shape) for a in om.grad_log_post(0, x[[0]])]
    for i in range(9):
        for a, gi in zip(ref, om.grad_log_post(int(lab[i]), x[[i]])):
            a += gi.astype(np.float64) ** 2
    ref = [a / 9 for a in ref]
    for a, b in zip(dev, ref):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-6 * np.abs(b).max())
    m.close()


def test_config5_loop_with_finetune_and_state(sess, tmp_path):
    """Config 5 proper at reduced size on the device: query round -> fine-tune on all labelled patches -> weights and
    queries on disk; the weights move between rounds (later posteriors differ from the loop without fine-tuning),
    two runs agree bit for bit, and a resumed run continues from the files."""
    from nnal_amd import al_loop, device, PW_AL
    torch = sess.torch
    ld, sk = netspec.net_c()
    in_shape = (16, 16, 16, 1)
    pars = netspec.he_init(ld, in_shape, seed=15, skips=sk)
    n, B, k = 600, 48, 8
    x = np.random.RandomState(1007).randn(n, 16 ** 3).astype(np.float32)
    labels = (x[:, :64].sum(1) > 0).astype(np.int64)
    pool = sess.to_device(x, torch.float32)

    def run(rounds, state_dir=None, ft=True):
        m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=64)
        m.set_weights(pars)
        m.get_optimizer(5e-3, [], 'SGD')
        kw = dict(labels=labels, finetune=dict(epochs=2, b=6), state=PW_AL.LoopState(state_dir) if state_dir else None) if ft else {}
        out = al_loop.run_rounds(m, sess, pool, rounds, B, k, seed=11, **kw)
        w = {nme: [a.copy() for a in wb] for nme, wb in m.var_dict.items()}
        m.close()
        return out, w

    d1 = str(tmp_path / 's1')
    r1, w1 = run(3, d1)
    r2, w2 = run(3)
    plain, w0 = run(3, ft=False)
    for a, b in zip(r1, r2):
        for key in ('queries', 'candidates', 'posts', 'A'):
            np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    for nme in w1:
        for a, b, c in zip(w1[nme], w2[nme], w0[nme]):
            np.testing.assert_array_equal(a, b)
        assert not np.array_equal(w1[nme][0], w0[nme][0]), nme                 # every layer's weights moved
    np.testing.assert_array_equal(r1[0]['queries'], plain[0]['queries'])
    assert not np.array_equal(r1[1]['posts'], plain[1]['posts'])
    assert all(len(r['finetune_loss']) > 0 and np.isfinite(r['finetune_loss']).all() for r in r1)
    for it in range(3):
        qm = np.loadtxt(os.path.join(d1, 'queries', '%d' % it), ndmin=2).astype(int)
        np.testing.assert_array_equal(qm[:, 0], r1[it]['queries'])
        f = np.load(os.path.join(d1, 'curr_weights_%d.npz' % (it + 1)))
        assert sorted(f.files) == sorted(nme + s_ for nme in w1 for s_ in ('/Weight', '/Bias'))
    d2 = str(tmp_path / 's2')
    run(2, d2)
    r3, w3 = run(3, d2)
    assert len(r3) == 1
    np.testing.assert_array_equal(r3[0]['queries'], r1[2]['queries'])
    for nme in w1:
        np.testing.assert_array_equal(w3[nme][0], w1[nme][0])


@pytest.mark.parametrize('tag,kind,feat,seed', [('neta', 'a', 3, 51), ('netb', 'bs', 7, 52), ('netc2d', 'c2', 4, 53), ('netc', 'c', 8, 54)])
def test_device_vs_reference_built_graph(sess, golden_dir, tag, kind, feat, seed):
    """The device against tests/golden/r2_refgraph.npz - values produced by the REFERENCE's own graph-construction code
    (NN.CNN / NN_extended.CNN, get_gradients, gen_A_matrices over tests/golden/tfshim.py): posteriors, feature layer (in the
    reference's flatten order), full gradient arrays of both classes and the A matrices."""
    from nnal_amd import PW_NNAL, device
    from tests.test_oracle_golden import refgraph_case
    g = np.load(os.path.join(golden_dir, 'r2_refgraph.npz'))
    ld, sk, in_shape, pars = refgraph_case(g, tag, kind, feat, seed)
    m = device.DeviceModel(sess, ld, in_shape, sk, feature_layer=feat, max_batch=8)
    m.set_weights(pars)
    x = g[tag + '_x']
    res = m.forward(x, want=('posteriors', 'feature_layer'))
    np.testing.assert_allclose(res['posteriors'], g[tag + '_post'], rtol=0, atol=2e-5)
    ref_feat = g[tag + '_feat']
    if ref_feat.ndim > 2:                   # NN_extended marks the un-flattened map: [N, ..., C] -> the flatten order [F, N]
        ref_feat = ref_feat.transpose(*reversed(range(ref_feat.ndim))).reshape(-1, ref_feat.shape[0])
    np.testing.assert_allclose(res['feature_layer'], ref_feat, rtol=0, atol=1e-5 * max(1., np.abs(ref_feat).max()))
    for j in (0, 1):
        dev = sess.run(m.grad_posts[str(j)], feed_dict={m.x: x[[2]], m.keep_prob: 1.})
        ref = [g['%s_grad%d_%d' % (tag, j, k)] for k in range(2 * m.L)]
        _close(dev, ref, name='%s class %d' % (tag, j))
    A = np.stack(PW_NNAL.gen_A_matrices(Expr({'patch_shape': in_shape[:3]}), m, sess, x, g[tag + '_post'][1].astype(np.float64), 1e-3))
    np.testing.assert_allclose(A, g[tag + '_A'], rtol=2e-3, atol=1e-7)
    m.close()


def test_edge_cases_of_the_round2_entry_points(sess):
    """Empty and degenerate inputs: zero-row lists, N = 0, a batch of one, unlabelled samples, keep_prob = 1 dropout."""
    from nnal_amd import device
    from nnal_amd._lib import AlqError, check
    import ctypes as C
    torch = sess.torch
    ld = netspec.net_a()
    in_shape = (20, 20, 1)
    m, pars = _mk(sess, ld, in_shape, (), 49, max_batch=4, dropout=[[0, 2], 0.5])
    x = sess.to_device(np.random.RandomState(12).randn(6, 400).astype(np.float32), torch.float32)
    empty = sess.empty((0,), torch.int64)
    post, _, _ = m.forward_device(x, 0, rows=empty)
    assert tuple(post.shape) == (2, 0)
    r = m.fisher_device(x, 0, None, 1e-3, rows=empty)
    assert tuple(r['A'].shape) == (0, 3, 3) and float(r['Asum'].abs().sum()) == 0.
    with pytest.raises(AlqError):                       # N = 0 gradients: nothing to differentiate
        m.param_grads_device(x, 0, 0)
    with pytest.raises(ValueError):                     # more than max_batch in one call
        m.param_grads_device(x, 6, 0)
    # keep_prob = 1: the dropout entry point is the plain forward pass, bit for bit
    p1, _ = m.forward_dropout_device(x, 6, 1.0, seed=5)
    p0, _, _ = m.forward_device(x, 6)
    assert torch.equal(p0, p1)
    # unlabelled samples (all-zero one-hot column) contribute nothing to the training gradient
    m.get_optimizer(0.01, [], 'SGD')
    om = OracleModel(ld, in_shape, pars)
    y = np.zeros((2, 3)); y[0, 0] = 1; y[1, 2] = 1      # sample 1 unlabelled
    g, _, loss = m.param_grads_device(x[:3], 3, 1, labels=np.array([0, -1, 1], np.int32), loss_scale=1. / 3, per_sample=False, want_loss=True)
    _, ref = om.loss_and_grads(x[:3].cpu().numpy().reshape(3, 20, 20, 1), y)
    _close(m.unflatten(g.cpu().numpy()), ref, name='unlabelled sample')
    with pytest.raises(AlqError):
        check(sess.lib.alq_sgd_step(sess.ctx, None, None, 5, 0.1))
    m.close()


def test_volume_level_experiment_on_the_device(sess, tmp_path):
    """PW_AL.Experiment_MultiImg.run_method (PW_AL.py:690-898) on synthetic NRRD subjects, everything on the device:
    grid indices -> query_multimg('fi') -> queries/<iter> -> finetune_multimg -> curr_weights_<iter>.  Two runs agree bit
    for bit; the files are what the loop reported; the reference's own 'PW' net (NN.create_model) runs the same loop."""
    from nnal_amd import NN, PW_AL
    from tests.test_dist_gloo import VOL_PARS, _subject_paths, _write_subjects
    data = str(tmp_path / 'data')
    os.makedirs(data)
    _write_subjects(data)

    def factory(e, in_shape, s):
        ld = netspec.net_a()
        m = NN.CNN(in_shape, ld, 'vol', len(ld) - 2, None, sess=s, max_batch=32)
        m.set_weights(netspec.he_init(ld, in_shape, seed=61, bias_std=0.05))
        m.get_optimizer(e.pars['learning_rate'], [], 'SGD')
        return m

    def run(root, method, max_q, fac=factory):
        expr = PW_AL.Experiment_MultiImg(str(tmp_path / root), VOL_PARS, _subject_paths(data))
        expr.model_factory = fac
        expr.add_method(method)
        np.random.seed(17)
        log = expr.run_method(method, max_q, sess=sess)
        w = {n: [a.copy() for a in wb] for n, wb in expr.model.var_dict.items()}
        expr.model.close()
        return log, w
    l1, w1 = run('e1', 'fi', 6)
    l2, w2 = run('e2', 'fi', 6)
    assert len(l1) >= 2 and sum(len(l['Q_mat']) for l in l1) >= 6
    for a, b in zip(l1, l2):
        np.testing.assert_array_equal(a['Q_mat'], b['Q_mat'])
    for n in w1:
        for a, b in zip(w1[n], w2[n]):
            np.testing.assert_array_equal(a, b)
    w0 = netspec.he_init(netspec.net_a(), (5, 5, 6), seed=61, bias_std=0.05)
    assert all(not np.array_equal(w1[n][0], w0[n][0]) for n in w1)           # the fine-tune moved every layer
    for it, l in enumerate(l1):
        f = np.loadtxt(os.path.join(str(tmp_path / 'e1'), 'fi', 'queries', '%d' % it), ndmin=2).astype(np.int64)
        np.testing.assert_array_equal(f, l['Q_mat'])
        assert os.path.exists(os.path.join(str(tmp_path / 'e1'), 'fi', 'AL_running_times', 'dt_%d' % it))
        assert os.path.exists(os.path.join(str(tmp_path / 'e1'), 'fi', 'curr_weights_%d.npz' % (it + 1)))
    # queried voxels are grid voxels of their subject with a mask label, none twice
    inds, _ = PW_AL.gen_multimg_inds(_subject_paths(data), VOL_PARS['grid_spacing'])
    allq = np.concatenate([l['Q_mat'] for l in l1])
    assert len(np.unique(allq, axis=0)) == len(allq)
    for v, s_ in allq:
        assert v in inds[s_]
    le, _ = run('e3', 'entropy', VOL_PARS['k'])
    assert len(le) == 1 and len(le[0]['Q_mat']) == VOL_PARS['k']
    # the reference's literal model: NN.create_model('PW', ...) = create_PW1 on the [5, 5, 6] patches of two modalities
    pars = dict(VOL_PARS, model_name='PW', learning_rate=1e-3)
    expr = PW_AL.Experiment_MultiImg(str(tmp_path / 'e4'), pars, _subject_paths(data))
    expr.add_method('fi')
    np.random.seed(18)
    lp = expr.run_method('fi', 1, sess=sess)
    assert len(lp) == 1 and len(lp[0]['Q_mat']) >= 1 and expr.model.L == 7
    expr.model.close()
