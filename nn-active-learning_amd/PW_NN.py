"""Batched patch evaluation (reference: PW_NN.py:357-539) on the device."""
import numpy as np

from . import patch_utils

_CHUNK = 8192   # indices gathered per device pass (results do not depend on it)


def mc_dropout_args(model, x_feed_dict):
    """(keep_prob, dropout active?, seed) of one batch_eval call: `x_feed_dict = {model.keep_prob: p}` overrides
    keep_prob = 1 (PW_NN.py:516-521); an evaluation with dropout draws ONE seed from the global NumPy stream (masks are
    then keyed by the sample's position).  A rank of a sharded evaluation that holds nothing of a subject calls this
    alone, so that every rank's stream advances alike."""
    keep_prob = 1.
    for k, val in x_feed_dict.items():
        if k is getattr(model, 'keep_prob', None):
            keep_prob = float(val)
    mc = keep_prob < 1. and len(model.dropout_layers) > 0
    seed = int(np.random.randint(0, 2 ** 31 - 1)) if mc else 0
    return keep_prob, mc, seed


def batch_eval(model, sess, img_dat, inds, patch_shape, batch_size, stats, varnames,
               mask=None, x_feed_dict={}, _vols=None, _first_sample=0):
    """PW_NN.batch_eval: evaluates `varnames` ('posteriors', 'prediction', 'feature_layer')
    of `model` on patches around voxels `inds` of the m padded modalities `img_dat`.

    Returns a list of float64 arrays: 'posteriors' -> [n] probability of class 1
    (PW_NN.py:526-529), 'prediction' -> [n], 'feature_layer' -> [fdim, n].  Patches are
    normalised with the channel-index rule of PW_NN.py:503-506 (channel j < m with stats[j]).
    `x_feed_dict = {model.keep_prob: p}` (PW_NN.py:516-521) evaluates with dropout on the model's dropout layers.
    `_first_sample` (not a reference argument): the position of inds[0] in the caller's whole evaluation - dropout masks
    are keyed by it, so an evaluation cut into blocks (pool_shard.work_block) draws the masks of the uncut one.
    `batch_size` only bounds the reference's feed size; samples are independent, so the device
    path walks the same index order in larger chunks.
    """
    if not isinstance(varnames, list):
        varnames = [varnames]
    for v in varnames:
        if v not in ('posteriors', 'prediction', 'feature_layer'):
            raise NotImplementedError("batch_eval variable %r (training-time graph) is outside the scored path" % v)
    keep_prob, mc, seed = mc_dropout_args(model, x_feed_dict)
    if mc and 'feature_layer' in varnames:
        raise NotImplementedError('feature_layer at keep_prob < 1')
    if not isinstance(img_dat[0], np.ndarray):
        # PW_NN.py:429-444: paths -> load (NRRD) and zero-pad by the patch radii
        from . import nrrd_io
        rads = [int((patch_shape[i] - 1) / 2.) for i in range(3)]
        img_dat = [np.pad(nrrd_io.read(p)[0], ((rads[0], rads[0]), (rads[1], rads[1]), (rads[2], rads[2])), 'constant')
                   for p in img_dat]
    if int(batch_size) < 1:
        raise ValueError('batch_size must be positive')
    m = len(img_dat)
    inds = np.asarray(inds)
    n = len(inds)
    # _vols (not a reference argument): volumes a caller of this package has already uploaded for the same query
    vols = _vols if _vols is not None else patch_utils.DeviceVolumes(sess, img_dat)
    want_pred = 'prediction' in varnames
    want_feat = 'feature_layer' in varnames
    posts = np.zeros(n)
    preds = np.zeros(n)
    feats = np.zeros((model.feature_dim, n)) if want_feat else None
    st = np.asarray(stats, dtype=np.float64)[:m]
    for a in range(0, n, _CHUNK):
        b = min(n, a + _CHUNK)
        t = vols.gather(inds[a:b], patch_shape, st, quirk=1)
        if mc:
            post, pred = model.forward_dropout_device(t, b - a, keep_prob, seed=seed, first_sample=_first_sample + a,
                                                      want_pred=want_pred)
            feat = None
        else:
            post, pred, feat = model.forward_device(t, b - a, want_pred, want_feat)
        posts[a:b] = post[1].cpu().numpy()
        if want_pred:
            preds[a:b] = pred.cpu().numpy()
        if want_feat:
            f = feat.cpu().numpy()
            if model._feature_perm is not None:
                f = f[:, model._feature_perm]
            feats[:, a:b] = f.T
    out = []
    for v in varnames:
        out.append({'posteriors': posts, 'prediction': preds, 'feature_layer': feats}[v])
    return out
