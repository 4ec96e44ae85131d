"""The oracle against outputs of the reference's own code (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference).  CPU only."""
import os

import numpy as np
import pytest

from oracle import alpath, netspec
from oracle.model import OracleModel, OracleSession


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


class Expr(object):
    def __init__(self, pars, train_stats=None, nclass=2):
        self.pars, self.train_stats, self.nclass = pars, train_stats, nclass


def test_shrink_gradient(golden_dir):
    g = _load(golden_dir, 'host_layers.npz')
    grads = [g['shrink_in_%d' % i] for i in range(6)]
    np.testing.assert_array_equal(alpath.shrink_gradient(grads), g['shrink_out'])


def test_global2local(golden_dir):
    g = _load(golden_dir, 'host_layers.npz')
    loc = alpath.global2local_inds(g['g2l_inds'], list(g['g2l_sizes']))
    for i, l in enumerate(loc):
        np.testing.assert_array_equal(l, g['g2l_out_%d' % i])


def test_binary_uncertainty_filter(golden_dir):
    g = _load(golden_dir, 'host_layers.npz')
    np.testing.assert_array_equal(alpath.binary_uncertainty_filter(g['buf_posts'], 37), g['buf_out'])


def test_entropy_guards(golden_dir):
    g = _load(golden_dir, 'host_layers.npz')
    a = g['ent_in'].copy()
    np.testing.assert_array_equal(alpath.compute_entropy(a), g['ent_out'])
    np.testing.assert_array_equal(a, g['ent_in_after'])          # in-place guard
    b = g['ent_in'].copy()
    np.testing.assert_array_equal(alpath.uncertainty_filtering(b, 11), g['uf_out'])
    np.testing.assert_array_equal(b, g['uf_in_after'])


def test_sample_query_dstr(golden_dir):
    g = _load(golden_dir, 'host_layers.npz')
    np.testing.assert_array_equal(alpath.sample_query_dstr(g['sq_q'], 25, g['sq_draws']), g['sq_out'])


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_get_patches(golden_dir, tag):
    g = _load(golden_dir, 'gather.npz')
    vols = [g[tag + '_vol0'], g[tag + '_vol1']]
    p, lab = alpath.get_patches(vols, g[tag + '_inds'], tuple(g[tag + '_pshape']), True, g[tag + '_mask'])
    np.testing.assert_array_equal(p, g[tag + '_patches'])
    np.testing.assert_array_equal(lab, g[tag + '_labels'])
    allimgs = [vols + [g[tag + '_mask']], vols[::-1] + [g[tag + '_mask']]]
    inds = g[tag + '_inds']
    P, Lb = alpath.get_patches_multimg(allimgs, [inds[:20], inds[20:]], tuple(g[tag + '_pshape']),
                                       g[tag + '_mm_stats'])
    for j in range(2):
        np.testing.assert_array_equal(P[j], g[tag + '_mm_p%d' % j])
        np.testing.assert_array_equal(Lb[j], g[tag + '_mm_l%d' % j])


def _neta_eval_setup(g):
    layer_dict = netspec.net_a()
    pshape = tuple(int(v) for v in g['pshape'])
    in_shape = (pshape[0], pshape[1], 2 * pshape[2])
    pars = netspec.he_init(layer_dict, in_shape, seed=int(g['wseed']), bias_std=0.05)
    model = OracleModel(layer_dict, in_shape, pars, feature_layer=len(layer_dict) - 2)
    return model, OracleSession(model), pshape


def test_batch_eval_and_entropy_query(golden_dir):
    g = _load(golden_dir, 'eval_neta.npz')
    model, sess, pshape = _neta_eval_setup(g)
    vols = [g['vol0'], g['vol1']]
    stats = g['stats'].tolist()
    r = alpath.batch_eval(model, sess, vols, g['pool'], pshape, 64, stats,
                          ['posteriors', 'prediction', 'feature_layer'])
    np.testing.assert_array_equal(r[0], g['be_posteriors'])
    np.testing.assert_array_equal(r[1], g['be_prediction'])
    np.testing.assert_array_equal(r[2], g['be_feature_layer'])
    expr = Expr({'patch_shape': pshape, 'ntb': 64, 'stats': stats, 'k': 20})
    np.testing.assert_array_equal(alpath.entropy_query(expr, model, sess, vols, g['pool']), g['entropy_q'])


def test_bin_uncertainty_filter_multimg(golden_dir):
    g = _load(golden_dir, 'eval_neta.npz')
    model, sess, pshape = _neta_eval_setup(g)
    vols = [g['vol0'], g['vol1']]
    mask = g['mask']
    allimgs = [vols + [mask], [vols[1], vols[0], mask]]
    pools = [g['pool'][:170], g['pool'][170:]]
    expr = Expr({'patch_shape': pshape, 'ntb': 50}, train_stats=g['mm_tstats'])
    sel_inds, sel_posts = alpath.bin_uncertainty_filter_multimg(expr, model, sess, allimgs, pools, 40)
    for j in range(2):
        np.testing.assert_array_equal(sel_inds[j], g['mm_sel_inds_%d' % j])
        np.testing.assert_array_equal(sel_posts[j], g['mm_sel_posts_%d' % j])


FISHER = [('fisher_neta.npz', 'a'), ('fisher_neta_saturated.npz', 'a'),
          ('fisher_netb_small_25x25x2.npz', 'bs'), ('fisher_netc2d.npz', 'c2'),
          ('fisher_netc_8cube.npz', 'c')]


def build_fisher_model(g, kind):
    """Rebuilds the model of a fisher_*.npz fixture from its seeds (shared with the GPU tests)."""
    in_shape = tuple(int(v) for v in g['in_shape'])
    skips = ()
    if kind == 'a':
        ld = netspec.net_a()
    elif kind == 'bs':
        ld = netspec.net_b_small()
    elif kind == 'b':
        ld = netspec.net_b()
    elif kind == 'c2':
        ld, skips = netspec.net_c_2d()
    else:
        ld, skips = netspec.net_c()
    pars = netspec.he_init(ld, in_shape, seed=int(g['wseed']), skips=skips, bias_std=float(g['bias_std']))
    ls = float(g['logit_scale'])
    if ls > 0:
        last = list(pars.keys())[-1]
        pars[last][0] = (pars[last][0] * ls).astype(np.float32)
        pars[last][1] = (pars[last][1] * ls).astype(np.float32)
        pars[last][1][1, 0] -= np.float32(float(g['logit_shift']) * ls)
    return ld, skips, in_shape, pars


@pytest.mark.parametrize('fname,kind', FISHER)
def test_gen_A_matrices(golden_dir, fname, kind):
    g = _load(golden_dir, fname)
    ld, skips, in_shape, pars = build_fisher_model(g, kind)
    model = OracleModel(ld, in_shape, pars, skips=skips)
    sess = OracleSession(model)
    n = min(int(g['n']), 12)
    x = g['x'][:n]
    p1 = model.forward(x)['posteriors'][1].astype(np.float64)
    np.testing.assert_array_equal(p1, g['p1'][:n])
    A = alpath.gen_A_matrices(Expr({'patch_shape': in_shape[:3]}), model, sess, x, p1,
                              float(g['diag_load']))
    np.testing.assert_array_equal(np.stack(A), g['A'][:n])


# ------------------------------------------------------------------------------------------ reference-built graphs
REFGRAPH = [('neta', 'a', 3, 51), ('netb', 'bs', 7, 52), ('netc2d', 'c2', 4, 53), ('netc', 'c', 8, 54)]


def refgraph_case(g, tag, kind, feat, seed):
    in_shape = tuple(int(v) for v in g[tag + '_in_shape'])
    skips = ()
    if kind == 'a':
        ld = netspec.net_a()
    elif kind == 'bs':
        ld = netspec.net_b_small()
    elif kind == 'c2':
        ld, skips = netspec.net_c_2d()
    else:
        ld, skips = netspec.net_c()
    pars = netspec.he_init(ld, in_shape, seed=seed, skips=skips, bias_std=0.05)
    return ld, skips, in_shape, pars


@pytest.mark.parametrize('tag,kind,feat,seed', REFGRAPH)
def test_oracle_graph_equals_reference_built_graph(golden_dir, tag, kind, feat, seed):
    """tests/golden/r2_refgraph.npz was produced by the REFERENCE's own NN.CNN / NN_extended.CNN constructors,
    get_gradients and gen_A_matrices running over a lazy-graph `tensorflow` stand-in (tests/golden/tfshim.py) whose op
    kernels are oracle.tfops: layer, variable, flatten and skip order and the gradient nodes are the reference's.
    The oracle's own graph must reproduce posteriors, the feature layer, every gradient array of log posteriors[j, 0]
    and the A matrices bit for bit."""
    g = _load(golden_dir, 'r2_refgraph.npz')
    ld, skips, in_shape, pars = refgraph_case(g, tag, kind, feat, seed)
    om = OracleModel(ld, in_shape, pars, skips=skips, feature_layer=feat)
    x = g[tag + '_x']
    o = om.forward(x)
    np.testing.assert_array_equal(o['posteriors'], g[tag + '_post'])
    np.testing.assert_array_equal(o['feature_layer'], g[tag + '_feat'])
    for j in (0, 1):
        grads = om.grad_log_post(j, x[[2]])
        for k, a in enumerate(grads):
            np.testing.assert_array_equal(a, g['%s_grad%d_%d' % (tag, j, k)])
        np.testing.assert_array_equal(alpath.shrink_gradient(grads), g['%s_shrunk%d' % (tag, j)])
    A = alpath.gen_A_matrices(Expr({'patch_shape': in_shape[:3]}), om, OracleSession(om), x, g[tag + '_post'][1].astype(np.float64), 1e-3)
    np.testing.assert_array_equal(np.stack(A), g[tag + '_A'])
