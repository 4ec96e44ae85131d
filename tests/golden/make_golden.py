#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own Python.

Run in the build container only (it needs /root/reference, which never travels):

    python tests/golden/make_golden.py

What is executed from the reference, unmodified: `NNAL_tools.shrink_gradient`,
`compute_entropy`, `uncertainty_filtering`, `sample_query_dstr`;
`patch_utils.get_patches`, `get_patches_multimg`, `global2local_inds`;
`PW_NNAL.binary_uncertainty_filter`, `gen_A_matrices`,
`bin_uncertainty_filter_multimg`, `CNN_query(...,'entropy')`; `PW_NN.batch_eval`.

What is NOT the reference: the TensorFlow graph.  TensorFlow (and h5py, nrrd,
skimage, cvxopt, cvxpy, the author's `alexnet`) are absent from the image, so
their names are registered as empty placeholder modules purely so that the
reference's `import` lines succeed; no attribute of them is ever called.  The
`sess`/`model` pair handed to the reference functions is
`oracle.model.OracleSession/OracleModel` (torch-CPU restatement of the graph).

The files hold data only: seeded inputs and the outputs the reference code
produced for them.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

ABSENT = ('tensorflow', 'h5py', 'nrrd', 'skimage', 'cvxopt', 'cvxpy', 'alexnet',
          'pydensecrf', 'nibabel', 'imageio', 'mosek')


class _Inert(type):
    """Metaclass of the inert names handed out by the placeholder modules: attribute access and
    item assignment succeed (module top levels do `solvers.options[...] = ...` and subclass
    `AlexNet`), nothing computes."""

    def __getattr__(cls, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Inert(name, (object,), {})

    def __setitem__(cls, key, value):
        pass


class _Placeholder(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Inert(name, (object,), {})


class _AbsentFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Placeholder(spec.name)

    def exec_module(self, module):
        pass


def import_reference():
    for name in ABSENT:
        try:
            __import__(name)
            raise SystemExit('%s is importable here: regenerate with the real module' % name)
        except ImportError:
            pass
    sys.meta_path.append(_AbsentFinder())
    sys.path.insert(0, REF)
    import NNAL_tools, patch_utils, PW_NN, PW_NNAL  # noqa: E401
    return NNAL_tools, patch_utils, PW_NN, PW_NNAL


class Expr(object):
    """The attributes of `PW_AL.Experiment` that the scored path reads."""

    def __init__(self, pars, train_stats=None, nclass=2):
        self.pars = pars
        self.nclass = nclass
        self.train_stats = train_stats


def main():
    from oracle import netspec
    from oracle.model import OracleModel, OracleSession
    NNAL_tools, patch_utils, PW_NN, PW_NNAL = import_reference()

    # ---------------------------------------------------------------- host layers
    rs = np.random.RandomState(101)
    shapes = [(5, 5, 1, 16), (16,), (3, 3, 16, 32), (32,), (2, 2048), (2, 1)]
    grads = [rs.randn(*s).astype(np.float32) * 1e-2 for s in shapes]
    host = {'shrink_in_%d' % i: g for i, g in enumerate(grads)}
    host['shrink_out'] = NNAL_tools.shrink_gradient(grads, 'sum')

    g2l_inds = np.array([7, 0, 3, 12, 4, 9, 5, 11])
    g2l_sizes = [4, 0, 5, 4]
    loc = patch_utils.global2local_inds(g2l_inds, g2l_sizes)
    host['g2l_inds'] = g2l_inds
    host['g2l_sizes'] = np.array(g2l_sizes)
    for i, l in enumerate(loc):
        host['g2l_out_%d' % i] = np.asarray(l)

    posts = rs.rand(500)
    host['buf_posts'] = posts
    host['buf_out'] = PW_NNAL.binary_uncertainty_filter(posts, 37)

    pm = rs.rand(2, 60).astype(np.float32)
    pm[0, ::7] = 0.
    pm[1] = 1. - pm[0]
    host['ent_in'] = pm.copy()
    p1 = pm.copy()
    host['ent_out'] = NNAL_tools.compute_entropy(p1)
    host['ent_in_after'] = p1
    p2 = pm.copy()
    host['uf_out'] = NNAL_tools.uncertainty_filtering(p2, 11)
    host['uf_in_after'] = p2

    q = rs.rand(200)
    q[::9] = -1e-3
    q /= q.sum()
    np.random.seed(77)
    host['sq_q'] = q.copy()
    host['sq_out'] = NNAL_tools.sample_query_dstr(q.copy(), 25, replacement=True)
    host['sq_draws'] = np.random.RandomState(77).random_sample(25)
    np.savez_compressed(os.path.join(HERE, 'host_layers.npz'), **host)

    # ---------------------------------------------------------------- gather
    rs = np.random.RandomState(7)
    gat = {}
    for tag, pshape in (('a', (5, 5, 3)), ('b', (7, 5, 1))):
        rad = [int((d - 1) / 2.) for d in pshape]
        orig = (40, 40, 12)
        vols = [np.pad(rs.randn(*orig), [(r, r) for r in rad], 'constant') for _ in range(2)]
        mask = (rs.rand(*orig) > .5).astype(np.int64)
        inds = rs.permutation(int(np.prod(orig)))[:40]
        # include the 8 corners: the whole window sits in the zero padding
        corners = [np.ravel_multi_index(c, orig) for c in
                   [(0, 0, 0), (39, 39, 11), (0, 39, 0), (39, 0, 11)]]
        inds = np.concatenate([inds, corners])
        p, lab = patch_utils.get_patches(vols, inds, pshape, True, mask)
        gat[tag + '_seed'] = np.array(7)
        gat[tag + '_pshape'] = np.array(pshape)
        gat[tag + '_vol0'] = vols[0]
        gat[tag + '_vol1'] = vols[1]
        gat[tag + '_mask'] = mask
        gat[tag + '_inds'] = inds
        gat[tag + '_patches'] = p
        gat[tag + '_labels'] = lab
        stats = np.array([[0.1, 1.3, -0.2, 0.7], [0.05, 0.9, 0.3, 1.1]])
        allimgs = [vols + [mask], vols[::-1] + [mask]]
        img_inds = [inds[:20], inds[20:]]
        P, Lb = patch_utils.get_patches_multimg(allimgs, img_inds, pshape, stats)
        gat[tag + '_mm_stats'] = stats
        for j in range(2):
            gat[tag + '_mm_p%d' % j] = P[j]
            gat[tag + '_mm_l%d' % j] = Lb[j]
    np.savez_compressed(os.path.join(HERE, 'gather.npz'), **gat)

    # ---------------------------------------------------------------- through-gather evaluation
    # NET-A on odd patches cut from a synthetic volume: pins batch_eval (with its channel-index
    # normalisation quirk, d3 = 3), CNN_query('entropy') and bin_uncertainty_filter_multimg.
    ev = {}
    rs = np.random.RandomState(21)
    pshape = (9, 9, 3)
    rad = [4, 4, 1]
    orig = (20, 18, 6)
    m = 2
    vols = [np.pad(rs.randn(*orig) * 1.5 + 0.3, [(r, r) for r in rad], 'constant') for _ in range(m)]
    mask = (rs.rand(*orig) > .5).astype(np.int64)
    layer_dict = netspec.net_a()
    in_shape = (9, 9, m * 3)
    pars = netspec.he_init(layer_dict, in_shape, seed=31, bias_std=0.05)
    model = OracleModel(layer_dict, in_shape, pars, feature_layer=len(layer_dict) - 2)
    sess = OracleSession(model)
    pool = rs.permutation(int(np.prod(orig)))[:300]
    stats = [[0.3, 1.5], [0.25, 1.4]]
    expr = Expr({'patch_shape': pshape, 'ntb': 64, 'stats': stats, 'k': 20})
    ev['orig_shape'] = np.array(orig)
    ev['pshape'] = np.array(pshape)
    ev['vol0'] = vols[0]
    ev['vol1'] = vols[1]
    ev['mask'] = mask
    ev['pool'] = pool
    ev['stats'] = np.array(stats)
    ev['wseed'] = np.array(31)
    r = PW_NN.batch_eval(model, sess, vols, pool, pshape, 64, stats,
                         ['posteriors', 'prediction', 'feature_layer'])
    ev['be_posteriors'] = r[0]
    ev['be_prediction'] = r[1]
    ev['be_feature_layer'] = r[2]
    ev['entropy_q'] = PW_NNAL.CNN_query(expr, model, sess, vols, pool, None, 'entropy')
    tstats = np.array([[0.3, 1.5, 0.25, 1.4], [0.2, 1.2, 0.35, 1.6]])
    expr2 = Expr({'patch_shape': pshape, 'ntb': 50}, train_stats=tstats)
    allimgs = [vols + [mask], [vols[1], vols[0], mask]]
    pools = [pool[:170], pool[170:]]
    sel_inds, sel_posts = PW_NNAL.bin_uncertainty_filter_multimg(
        expr2, model, sess, allimgs, pools, 40)
    ev['mm_tstats'] = tstats
    for j in range(2):
        ev['mm_sel_inds_%d' % j] = np.asarray(sel_inds[j])
        ev['mm_sel_posts_%d' % j] = np.asarray(sel_posts[j])
    np.savez_compressed(os.path.join(HERE, 'eval_neta.npz'), **ev)

    # ---------------------------------------------------------------- Fisher (gen_A_matrices)
    def fisher_case(fname, layer_dict, in_shape, wseed, xseed, n, skips=(), diag_load=1e-5,
                    bias_std=0.05, logit_scale=None, logit_shift=0., x_scale=1.0):
        pars = netspec.he_init(layer_dict, in_shape, seed=wseed, skips=skips, bias_std=bias_std)
        if logit_scale is not None:
            last = list(pars.keys())[-1]
            pars[last][0] = (pars[last][0] * logit_scale).astype(np.float32)
            pars[last][1] = (pars[last][1] * logit_scale).astype(np.float32)
            pars[last][1][1, 0] -= np.float32(logit_shift * logit_scale)
        mdl = OracleModel(layer_dict, in_shape, pars, skips=skips)
        ss = OracleSession(mdl)
        x = (np.random.RandomState(xseed).randn(n, *in_shape) * x_scale).astype(np.float32)
        post = mdl.forward(x)['posteriors'][1].astype(np.float64)   # what batch_eval stores
        expr = Expr({'patch_shape': in_shape[:3]})
        A = PW_NNAL.gen_A_matrices(expr, mdl, ss, x, post, diag_load)
        d = {'wseed': np.array(wseed), 'xseed': np.array(xseed), 'n': np.array(n),
             'in_shape': np.array(in_shape), 'bias_std': np.array(bias_std),
             'diag_load': np.array(diag_load), 'x_scale': np.array(x_scale),
             'logit_scale': np.array(-1. if logit_scale is None else logit_scale),
             'logit_shift': np.array(logit_shift),
             'x': x, 'p1': post, 'A': np.stack(A)}
        # both shrunk gradients through the reference's own shrink_gradient
        g0 = np.zeros((n, mdl.nlayers_par))
        g1 = np.zeros((n, mdl.nlayers_par))
        for i in range(n):
            feed = {mdl.x: x[i:i + 1], mdl.keep_prob: 1.}
            g0[i] = NNAL_tools.shrink_gradient(ss.run(mdl.grad_posts['0'], feed), 'sum')
            g1[i] = NNAL_tools.shrink_gradient(ss.run(mdl.grad_posts['1'], feed), 'sum')
        d['g0'] = g0
        d['g1'] = g1
        np.savez_compressed(os.path.join(HERE, fname), **d)
        print(fname, 'p1 range', post.min(), post.max(),
              'branches', int((post < 1e-6).sum()), int((post > 1 - 1e-6).sum()))

    fisher_case('fisher_neta.npz', netspec.net_a(), (32, 32, 1), 12, 1002, 48)
    fisher_case('fisher_neta_saturated.npz', netspec.net_a(), (32, 32, 1), 12, 1002, 48,
                logit_scale=8., logit_shift=3.)
    fisher_case('fisher_netb_small_25x25x2.npz', netspec.net_b_small(), (25, 25, 2), 13, 1003, 12,
                diag_load=1e-3)
    fisher_case('fisher_netb_25x25x2.npz', netspec.net_b(), (25, 25, 2), 13, 1003, 6,
                diag_load=1e-3)
    lc, sk = netspec.net_c_2d()
    fisher_case('fisher_netc2d.npz', lc, (16, 16, 1), 16, 1006, 16, skips=sk, diag_load=1e-3)
    lc, sk = netspec.net_c()
    fisher_case('fisher_netc_8cube.npz', lc, (8, 8, 8, 1), 14, 1004, 12, skips=sk, diag_load=1e-3)
    fisher_case('fisher_netc_32cube.npz', lc, (32, 32, 32, 1), 14, 1004, 3, skips=sk,
                diag_load=1e-3)


if __name__ == '__main__':
    main()
