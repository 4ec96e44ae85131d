#!/usr/bin/env python3
"""Round-3 goldens, produced by running the REFERENCE's own Python in the build container:

    python tests/golden/make_golden_r3.py   ->  tests/golden/r3_*.npz

Executed from /root/reference, unmodified (absent third-party names are inert placeholders, see make_golden.py):

  * PW_NNAL.gen_A_matrices + NNAL_tools.shrink_gradient on NET-B = NN.create_PW1's layer dict at the BENCHMARKED input
    shape [N, 32, 32, 32] (42 M parameters, weights regenerated from seed 13, never stored;
    the patches are regenerated from their seed too)                                   -> r3_fisher_netb_32ch.npz
  * PW_NNAL.query_multimg(..., 'rep-entropy') and (..., 'core-set') (PW_NNAL.py:284-351, :353-451) against the fake
    `sess` / `model` of the oracle                                                     -> r3_strategies.npz
  * PW_AL.finetune and PW_AL.finetune_multimg (PW_AL.py:1030-1147) against a RECORDING `sess`: the feeds of every
    `sess.run(model.train_step, ...)` (patches, one-hot labels, keep_prob) in call order -> r3_finetune.npz
  * NNAL.CNN_query(model, expr, pool_inds, 'fi', session) (NNAL.py:312-464), the image-level multi-class Fisher query,
    for c = 3 and c = 12 classes (the second takes the "ten largest posteriors" branch, :381-394)
                                                                                        -> r3_imgfi.npz
    Stand-ins bound for that run: `NN.cv2` (the reference's `import cv2` is commented out, NN.py:9, so `load_winds`
    cannot run as it stands) = an in-memory image table with an identity `resize`; `NNAL_tools.SDP_query_distribution`
    = a recorder that stores its arguments (the A list, lambda, the refined + centred feature matrix, k) and returns the
    uniform distribution (cvxopt is absent: the solver's iterate is unpinned, make_golden_r2.py pins its statement).
  * PW_AL.Experiment_MultiImg.run_method's pool bookkeeping between query and fine-tune (PW_AL.py:857-884), run on the
    reference's own lines through a small harness is NOT attempted: it is interleaved with TensorFlow session code.

The files hold data only."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
from make_golden import Expr  # noqa: E402


def fisher_netb_32ch(PW_NNAL, NNAL_tools):
    from oracle import netspec
    from oracle.model import OracleModel, OracleSession
    ld = netspec.net_b()
    in_shape = (32, 32, 32)
    wseed, xseed, n, diag_load, bias_std = 13, 1003, 8, 1e-3, 0.05
    pars = netspec.he_init(ld, in_shape, seed=wseed, bias_std=bias_std)
    mdl = OracleModel(ld, in_shape, pars)
    ss = OracleSession(mdl)
    x = np.random.RandomState(xseed).randn(n, *in_shape).astype(np.float32)
    post = mdl.forward(x)['posteriors'][1].astype(np.float64)
    A = PW_NNAL.gen_A_matrices(Expr({'patch_shape': in_shape}), mdl, ss, x, post, diag_load)
    g0 = np.zeros((n, mdl.nlayers_par))
    g1 = np.zeros((n, mdl.nlayers_par))
    for i in range(n):
        feed = {mdl.x: x[i:i + 1], mdl.keep_prob: 1.}
        g0[i] = NNAL_tools.shrink_gradient(ss.run(mdl.grad_posts['0'], feed), 'sum')
        g1[i] = NNAL_tools.shrink_gradient(ss.run(mdl.grad_posts['1'], feed), 'sum')
    d = dict(wseed=np.array(wseed), xseed=np.array(xseed), n=np.array(n), in_shape=np.array(in_shape),
             bias_std=np.array(bias_std), diag_load=np.array(diag_load), p1=post, A=np.stack(A), g0=g0, g1=g1,
             x_head=x[:, :2, :2, :4].copy())          # a corner of the regenerated patches, to check the regeneration
    np.savez_compressed(os.path.join(HERE, 'r3_fisher_netb_32ch.npz'), **d)
    print('r3_fisher_netb_32ch: p1', post.min(), post.max())


def small_subjects(rs, orig_shapes, rad, m=2):
    subs = []
    for shp in orig_shapes:
        vols = [np.pad(rs.randn(*shp) * 1.5 + 0.3, [(r, r) for r in rad], 'constant') for _ in range(m)]
        mask = (rs.rand(*shp) > .5).astype(np.int64)
        subs.append(vols + [mask])
    return subs


def strategies(PW_NNAL):
    from oracle import netspec
    from oracle.model import OracleModel, OracleSession
    out = {}
    rs = np.random.RandomState(3101)
    pshape, rad, m = (9, 9, 3), [4, 4, 1], 2
    shapes = [(14, 12, 5), (12, 13, 4)]
    subs = small_subjects(rs, shapes, rad, m)
    ld = netspec.net_a()
    in_shape = (9, 9, m * 3)
    pars = netspec.he_init(ld, in_shape, seed=41, bias_std=0.05)
    model = OracleModel(ld, in_shape, pars, feature_layer=len(ld) - 2)
    sess = OracleSession(model)
    pools = [rs.permutation(int(np.prod(shapes[0])))[:90], rs.permutation(int(np.prod(shapes[1])))[:70]]
    labeled = [rs.permutation(int(np.prod(shapes[0])))[:15], rs.permutation(int(np.prod(shapes[1])))[:25]]
    tstats = np.array([[0.3, 1.5, 0.25, 1.4], [0.2, 1.2, 0.35, 1.6]])
    out.update(pshape=np.array(pshape), wseed=np.array(41), tstats=tstats)
    for s_, sub in enumerate(subs):
        for j, v in enumerate(sub):
            out['sub%d_%d' % (s_, j)] = v
        out['pool_%d' % s_] = pools[s_]
        out['labeled_%d' % s_] = labeled[s_]
    for tag, B, k in (('re_a', 24, 6), ('re_b', 40, 9)):
        expr = Expr({'patch_shape': pshape, 'ntb': 37, 'k': k, 'B': B}, train_stats=tstats)
        Q = PW_NNAL.query_multimg(expr, model, sess, subs, [list(p) for p in pools], [list(l) for l in labeled], 'rep-entropy')
        out[tag + '_Bk'] = np.array([B, k])
        for s_ in range(2):
            out['%s_Q_%d' % (tag, s_)] = np.asarray(Q[s_], dtype=np.int64)
    for tag, k, seed in (('cs_a', 7, 5), ('cs_b', 12, 6)):
        expr = Expr({'patch_shape': pshape, 'ntb': 37, 'k': k, 'B': 24}, train_stats=tstats)
        expr.labeled_stats = tstats
        expr.train_paths = expr.labeled_paths = ['same']
        expr.root_dir = '/nonexistent'
        np.random.seed(seed)               # NN.gen_batch_inds(nT, 1000) permutes the labelled voxels of the last subject
        Q = PW_NNAL.query_multimg(expr, model, sess, subs, [list(p) for p in pools], [list(l) for l in labeled], 'core-set')
        out[tag + '_k_seed'] = np.array([k, seed])
        for s_ in range(2):
            out['%s_Q_%d' % (tag, s_)] = np.asarray(Q[s_], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'r3_strategies.npz'), **out)
    print('r3_strategies:', {k: v.tolist() for k, v in out.items() if '_Q_' in k})


class RecordingSession(object):
    def __init__(self):
        self.feeds = []

    def run(self, fetch, feed_dict=None):
        assert fetch == 'train_step'
        self.feeds.append(feed_dict)


class TrainModel(object):
    """The attributes PW_AL.finetune* read from the model (PW_AL.py:1075-1080, :1140-1146)."""
    train_step, x, y_, keep_prob = 'train_step', 'x', 'y_', 'keep_prob'
    dropout_rate = 0.5


def finetune_feeds(PW_AL):
    out = {}
    rs = np.random.RandomState(3201)
    pshape, rad, m = (5, 5, 3), [2, 2, 1], 2
    shapes = [(12, 10, 6), (9, 11, 5)]
    subs = small_subjects(rs, shapes, rad, m)
    for s_, sub in enumerate(subs):
        for j, v in enumerate(sub):
            out['sub%d_%d' % (s_, j)] = v
    out['pshape'] = np.array(pshape)
    model = TrainModel()

    def pack(tag, feeds):
        out[tag + '_lens'] = np.array([f['x'].shape[0] for f in feeds])
        out[tag + '_x'] = np.concatenate([f['x'] for f in feeds])
        out[tag + '_y'] = np.concatenate([f['y_'] for f in feeds], axis=1)
        out[tag + '_kp'] = np.array([f['keep_prob'] for f in feeds])

    # single image (PW_AL.py:1030-1088): channel-index normalisation, stats from expr.pars
    stats = [[0.3, 1.5], [0.25, 1.4]]
    train_inds = rs.permutation(int(np.prod(shapes[0])))[:23]
    expr = Expr({'patch_shape': pshape, 'b': 5, 'epochs': 2, 'stats': stats})
    sess = RecordingSession()
    np.random.seed(91)
    PW_AL.finetune(model, sess, expr, subs[0][:m], subs[0][m], train_inds)
    out.update(ft_train_inds=train_inds, ft_stats=np.array(stats), ft_seed=np.array(91), ft_b_epochs=np.array([5, 2]))
    pack('ft', sess.feeds)

    # multi image (PW_AL.py:1091-1147): slab normalisation with expr.train_stats; one subject with few voxels so that
    # some batches hold nothing of it
    tstats = np.array([[0.3, 1.5, 0.25, 1.4], [0.2, 1.2, 0.35, 1.6]])
    tr = [list(rs.permutation(int(np.prod(shapes[0])))[:17]), list(rs.permutation(int(np.prod(shapes[1])))[:3])]
    expr = Expr({'patch_shape': pshape, 'b': 4, 'epochs': 2}, train_stats=tstats)
    sess = RecordingSession()
    np.random.seed(92)
    PW_AL.finetune_multimg(expr, model, sess, subs, tr)
    out.update(fm_tstats=tstats, fm_seed=np.array(92), fm_b_epochs=np.array([4, 2]))
    for s_ in range(2):
        out['fm_train_inds_%d' % s_] = np.array(tr[s_], dtype=np.int64)
    pack('fm', sess.feeds)
    np.savez_compressed(os.path.join(HERE, 'r3_finetune.npz'), **out)
    print('r3_finetune: %d + %d recorded steps' % (len(out['ft_lens']), len(out['fm_lens'])))


def image_level_fi(NNAL_tools):
    """NNAL.CNN_query(..., 'fi') (NNAL.py:312-464) on a NET-A-shaped image classifier."""
    import NN
    import NNAL
    from oracle import netspec
    from oracle.model import OracleModel, OracleSession

    table = {}

    class CV2(object):
        @staticmethod
        def imread(path):
            return table[path].copy()          # cv2.imread hands out a fresh array per call (load_winds subtracts in place)

        @staticmethod
        def resize(img, target_shape):
            assert tuple(img.shape[:2]) == tuple(target_shape)
            return img
    NN.cv2 = CV2

    class Shape(object):
        def __init__(self, dims):
            self.dims = dims

        def __getitem__(self, i):
            return type('D', (), {'value': self.dims[i]})()

    class TFTensor(object):
        def __init__(self, handle, dims):
            self.handle, self.dims = handle, dims

        def get_shape(self):
            return Shape(self.dims)

    class ImgModel(object):
        """The attributes NNAL.CNN_query / NN.CNN.extract_features read (NNAL.py:312-464, NN.py:522-554)."""

        def __init__(self, om):
            self.om = om
            self.x, self.keep_prob = om.x, om.keep_prob
            self.posteriors = om.posteriors
            self.output = TFTensor(None, (om.nclass, None))
            self.feature_layer = TFTensor(om.feature_layer, (om.feature_layer.shape[0].value,))
            self.grad_posts = om.grad_posts

        def extract_features(self, inds, expr, session):
            return NN.CNN.extract_features(self, inds, expr, session)      # the reference's method body, verbatim

    class ImgSession(OracleSession):
        def run(self, fetch, feed_dict=None):
            if isinstance(fetch, dict):                                      # NNAL.py:383,397: dict of gradient lists
                return {k: OracleSession.run(self, v, feed_dict) for k, v in fetch.items()}
            if isinstance(fetch, TFTensor):
                fetch = fetch.handle
            return OracleSession.run(self, fetch, feed_dict)

    rec = {}

    def sdp_recorder(A, lambda_, X_pool, k):
        rec.update(A=np.stack(A), lambda_=lambda_, F=np.array(X_pool), k=k)
        n = len(A)
        return {'status': 'recorded', 'x': np.ones(n + A[0].shape[0]) / n}
    NNAL_tools.SDP_query_distribution = sdp_recorder

    out = {}
    rs = np.random.RandomState(3301)
    npool, hw = 48, 16
    imgs = rs.randn(npool, hw, hw, 3) + 100.
    tmp = tempfile.mkdtemp()
    pfile = os.path.join(tmp, 'paths.txt')
    with open(pfile, 'w') as f:
        for i in range(npool):
            table['img_%d' % i] = imgs[i]
            f.write('img_%d\n' % i)
    out['imgs'] = imgs
    for tag, c, wseed, lscale in (('c3', 3, 51, 6.), ('c12', 12, 52, 1.)):
        ld = netspec.net_a(nclass=c)
        in_shape = (hw, hw, 3)
        pars = netspec.he_init(ld, in_shape, seed=wseed, bias_std=0.05)
        last = list(pars.keys())[-1]
        # logits scaled so that class posteriors spread over several decades (c3: some below the 1e-6 cut-off)
        pars[last][0] = (pars[last][0] * lscale).astype(np.float32)
        om = OracleModel(ld, in_shape, pars, feature_layer=len(ld) - 2)
        model = ImgModel(om)
        sess = ImgSession(om)
        expr = Expr({'k': 5, 'B': 14, 'lambda_': 0.5, 'batch_size': 8, 'target_shape': (hw, hw), 'mean': 100.})
        expr.imgs_path_file = pfile
        pool_inds = rs.permutation(npool)[:40]
        np.random.seed(60 + c)
        rec.clear()
        Q = NNAL.CNN_query(model, expr, pool_inds, 'fi', sess, col=True)
        out.update({tag + '_pool_inds': pool_inds, tag + '_Q': np.asarray(Q), tag + '_A': rec['A'], tag + '_F': rec['F'],
                    tag + '_meta': np.array([c, wseed, 60 + c, expr.pars['k'], expr.pars['B']]),
                    tag + '_logit_scale': np.array(lscale)})
        post = om.forward(np.stack([table['img_%d' % i] - 100. for i in pool_inds]))['posteriors']
        print('r3_imgfi %s: Q' % tag, np.asarray(Q).tolist(), 'posteriors below 1e-6:', int((post < 1e-6).sum()),
              'classes >= 1e-6 per sample (min, max):', int((post >= 1e-6).sum(0).min()), int((post >= 1e-6).sum(0).max()))
    np.savez_compressed(os.path.join(HERE, 'r3_imgfi.npz'), **out)


def main():
    NNAL_tools, patch_utils, PW_NN, PW_NNAL = make_golden.import_reference()
    import PW_AL
    which = set(sys.argv[1:]) or {'netb', 'strategies', 'finetune', 'imgfi'}
    if 'strategies' in which:
        strategies(PW_NNAL)
    if 'finetune' in which:
        finetune_feeds(PW_AL)
    if 'imgfi' in which:
        image_level_fi(NNAL_tools)
    if 'netb' in which:
        fisher_netb_32ch(PW_NNAL, NNAL_tools)


if __name__ == '__main__':
    main()
