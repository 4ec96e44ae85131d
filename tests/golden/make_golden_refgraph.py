#!/usr/bin/env python3
"""Goldens from the reference's OWN graph-construction code (build container only):

    python tests/golden/make_golden_refgraph.py      ->  tests/golden/r2_refgraph.npz

`tests/golden/tfshim.py` is registered as `tensorflow`; then, unmodified from /root/reference:
  * NN.CNN(x, layer_dict, name, feature_layer, dropout) + get_optimizer + get_gradients      (NN.py:56-645)
  * NN_extended.CNN(x, layer_dict, name, skips, feature_layer) + get_gradients                (NN_extended.py:65-601, 1011-1035)
  * PW_NNAL.gen_A_matrices, NNAL_tools.shrink_gradient on `sess.run(model.grad_posts[j])`      (PW_NNAL.py:738-816)
are executed for NET-A, NET-B (narrow fc), NET-C 2-D and NET-C 3-D.  Layer / variable / flatten / skip order and the
gradient nodes are therefore the reference's; the op kernels are oracle.tfops (torch-CPU), "parity unpinned".
The script asserts that the oracle's own graph (oracle.model.OracleModel) gives the SAME BITS for posteriors, the
feature layer, every gradient array and the A matrices, and stores the values for the CPU / GPU tests."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
import tfshim  # noqa: E402


def import_reference_with_shim():
    tfshim.__path__ = []
    sys.modules['tensorflow'] = tfshim
    make_golden.ABSENT = tuple(n for n in make_golden.ABSENT if n != 'tensorflow')

    class SubFinder(make_golden._AbsentFinder):
        def find_spec(self, fullname, path, target=None):
            if fullname.startswith('tensorflow.'):
                import importlib.machinery
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None
    sys.meta_path.append(SubFinder())
    mods = make_golden.import_reference()
    import NN
    import NN_extended
    return mods + (NN, NN_extended)


def main():
    from oracle import netspec
    from oracle.model import OracleModel
    NNAL_tools, patch_utils, PW_NN, PW_NNAL, NN, NN_extended = import_reference_with_shim()
    tf = tfshim
    out = {}
    ld_c, sk_c = netspec.net_c()
    ld_c2, sk_c2 = netspec.net_c_2d()
    cases = [('neta', 'NN', netspec.net_a(), (20, 20, 1), (), 3, 51),
             ('netb', 'NN', netspec.net_b_small(), (25, 25, 2), (), 7, 52),
             ('netc2d', 'EXT', ld_c2, (16, 16, 2), sk_c2, 4, 53),
             ('netc', 'EXT', ld_c, (8, 8, 8, 1), sk_c, 8, 54)]
    for tag, schema, ld, in_shape, skips, feat, seed in cases:
        tf.reset_default_graph()
        pars = netspec.he_init(ld, in_shape, seed=seed, skips=skips, bias_std=0.05)
        x = tf.placeholder(tf.float32, [None] + list(in_shape), name='input')
        # the reference mutates the layer lists (NN_extended.py:213-214): hand it a copy
        ldc = type(ld)((k, [list(v) if isinstance(v, list) else v for v in spec] if False else list(spec)) for k, spec in ld.items())
        if schema == 'NN':
            model = NN.CNN(x, ldc, tag, feat, [[len(ld) - 1], 1.], [])
            model.get_optimizer(1e-3, [], 'SGD')
            model.get_gradients()
        else:
            model = NN_extended.CNN(x, ldc, tag, [list(s) for s in skips], feat, None)
            model.get_gradients()
        names = list(ld.keys())
        pnames = [n for n in names if n in pars]
        assert [n for n in names if n in model.var_dict] == pnames, 'parameterised layers differ'
        # variable creation order must be [W1, b1, W2, b2, ...] of the layer order (what shrink_gradient pairs up)
        tv = tf.trainable_variables()
        flat = [v for n in pnames for v in model.var_dict[n][-2:]]
        assert len(tv) == len(flat) and all(a is b for a, b in zip(tv, flat)), 'trainable_variables order'
        for n in pnames:
            W, b = model.var_dict[n][-2:]
            assert tuple(W.tensor.shape) == tuple(pars[n][0].shape) and W.tensor.numel() and b.tensor.numel() == pars[n][1].size
            W.load(pars[n][0])
            b.load(pars[n][1])
        sess = tf.Session()
        rs = np.random.RandomState(seed + 100)
        xs = rs.randn(6, *in_shape).astype(np.float32)
        feed = {model.x: xs, model.keep_prob: 1.}
        post = sess.run(model.posteriors, feed_dict=feed)
        featv = sess.run(model.feature_layer, feed_dict=feed)
        pred = sess.run(model.prediction, feed_dict=feed) if hasattr(model, 'prediction') else post.argmax(0)
        om = OracleModel(ld, in_shape, pars, skips=skips, feature_layer=feat)
        o = om.forward(xs)
        np.testing.assert_array_equal(post, o['posteriors'], err_msg=tag + ' posteriors')
        np.testing.assert_array_equal(np.asarray(featv), o['feature_layer'], err_msg=tag + ' feature layer')
        np.testing.assert_array_equal(pred, o['prediction'])
        out[tag + '_x'] = xs
        out[tag + '_post'] = post
        out[tag + '_feat'] = np.asarray(featv)
        out[tag + '_seed'] = np.int64(seed)
        out[tag + '_in_shape'] = np.array(in_shape)
        for j in (0, 1):
            g_ref = sess.run(model.grad_posts[str(j)], feed_dict={model.x: xs[[2]], model.keep_prob: 1.})
            g_orc = om.grad_log_post(j, xs[[2]])
            assert len(g_ref) == len(g_orc)
            for k, (a, b) in enumerate(zip(g_ref, g_orc)):
                np.testing.assert_array_equal(a, b, err_msg='%s grad class %d array %d' % (tag, j, k))
                out['%s_grad%d_%d' % (tag, j, k)] = a
            out['%s_shrunk%d' % (tag, j)] = NNAL_tools.shrink_gradient(g_ref, 'sum')
        expr = make_golden.Expr({'patch_shape': in_shape[:3]})
        p1 = post[1].astype(np.float64)
        A = np.stack(PW_NNAL.gen_A_matrices(expr, model, sess, xs, p1, 1e-3))
        from oracle import alpath
        from oracle.model import OracleSession
        A_orc = np.stack(alpath.gen_A_matrices(expr, om, OracleSession(om), xs, p1, 1e-3))
        np.testing.assert_array_equal(A, A_orc, err_msg=tag + ' A matrices')
        out[tag + '_A'] = A
        print('%-7s reference graph == oracle graph, bit for bit: posteriors %s, features %s, %d gradient arrays x 2, A %s' %
              (tag, post.shape, np.asarray(featv).shape, len(g_ref), A.shape))
    np.savez_compressed(os.path.join(HERE, 'r2_refgraph.npz'), **out)
    print('wrote r2_refgraph.npz (%d arrays)' % len(out))


if __name__ == '__main__':
    main()
