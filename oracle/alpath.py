"""Oracle restatement of the host-side functions of the query-scoring path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function names the
reference lines it restates; `tests/test_oracle_golden.py` checks every one of
them against outputs of the reference's own code (tests/golden/*.npz).
"""
import numpy as np


# ---------------------------------------------------------------- patch gather
def patch_radii(patch_shape):
    """`int((d-1)/2.)` per dim: patch_utils.py:1119-1121 / PW_NN.py:424-426."""
    return [int((d - 1) / 2.) for d in patch_shape]


def get_patches(imgs, inds, patch_shape, padded=True, mask=None):
    """patch_utils.get_patches (patch_utils.py:1087-1173).

    `imgs`: m volumes (already zero-padded by the radii when `padded`); `inds`:
    raveled C-order voxel indices in the UN-padded shape; result float64
    [b, d1, d2, m*d3] with modality j in channels j*d3:(j+1)*d3."""
    d1, d2, d3 = patch_shape
    r = patch_radii(patch_shape)
    vols = [np.asarray(v) for v in imgs]
    if not padded:
        vols = [np.pad(v, [(r[0], r[0]), (r[1], r[1]), (r[2], r[2])], 'constant') for v in vols]
    pshape = vols[0].shape
    orig = tuple(pshape[a] - 2 * r[a] for a in range(3))
    zi, yi, xi = np.unravel_index(np.asarray(inds), orig)
    b = len(inds)
    out = np.zeros((b, d1, d2, len(vols) * d3))
    for i in range(b):
        # centre in padded coordinates is index + radius, window is centre +- radius
        z0, y0, x0 = zi[i], yi[i], xi[i]
        for j, v in enumerate(vols):
            out[i, :, :, j * d3:(j + 1) * d3] = v[z0:z0 + 2 * r[0] + 1,
                                                  y0:y0 + 2 * r[1] + 1,
                                                  x0:x0 + 2 * r[2] + 1]
    if mask is not None:
        return out, np.asarray(mask)[(zi, yi, xi)]
    return out


def get_patches_multimg(all_padded_imgs, img_inds, patch_shape, stats):
    """patch_utils.get_patches_multimg (patch_utils.py:1175-1212): per-subject gather,
    SLAB normalisation (k*d3:(k+1)*d3) with stats[j, 2k], stats[j, 2k+1]."""
    m = len(all_padded_imgs[0]) - 1
    d3 = patch_shape[2]
    P = [[] for _ in img_inds]
    Lb = [[] for _ in img_inds]
    for j, inds in enumerate(img_inds):
        if len(inds) == 0:
            continue
        p, lab = get_patches(all_padded_imgs[j][:m], inds, patch_shape, True,
                             all_padded_imgs[j][m])
        for k in range(m):
            sl = slice(k * d3, (k + 1) * d3)
            p[:, :, :, sl] = (p[:, :, :, sl] - stats[j, 2 * k]) / stats[j, 2 * k + 1]
        P[j] = p
        Lb[j] = lab
    return P, Lb


def normalise_channels_quirk(patches, stats):
    """The in-place normalisation of batch_eval / CNN_query: CHANNEL j for j < m, not the
    depth slab (PW_NN.py:503-506, PW_NNAL.py:125-129)."""
    for j in range(len(stats)):
        patches[:, :, :, j] = (patches[:, :, :, j] - stats[j][0]) / stats[j][1]
    return patches


def global2local_inds(batch_inds, set_sizes):
    """patch_utils.global2local_inds (patch_utils.py:829-866): split global positions over
    concatenated sets, keeping the input order inside each set."""
    batch_inds = np.asarray(batch_inds)
    ends = np.cumsum(set_sizes)
    starts = ends - np.asarray(set_sizes)
    owner = np.searchsorted(ends, batch_inds, side='right')
    return [batch_inds[owner == s] - starts[s] for s in range(len(set_sizes))]


# ---------------------------------------------------------------- evaluation
def batch_eval(model, sess, img_dat, inds, patch_shape, batch_size, stats, varnames,
               mask=None, x_feed_dict={}):
    """PW_NN.batch_eval (PW_NN.py:357-539) for 'posteriors' / 'prediction' / 'feature_layer'."""
    if not isinstance(varnames, list):
        varnames = [varnames]
    m = len(img_dat)
    n = len(inds)
    inds = np.asarray(inds)
    cuts = list(range(0, n, batch_size)) + [n]          # PW_NN.py:447-451
    results = []
    for var in varnames:
        if var == 'feature_layer':
            vals = np.zeros((model.feature_layer.shape[0].value, n))
        else:
            vals = np.zeros(n)
        handle = getattr(model, var)
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b <= a:
                continue
            t = get_patches(img_dat, inds[a:b], patch_shape)
            normalise_channels_quirk(t, stats[:m])
            feed = {model.x: t, model.keep_prob: 1.}
            feed.update(x_feed_dict)
            bv = sess.run(handle, feed_dict=feed)
            if var == 'posteriors':
                vals[a:b] = bv[1, :]                     # PW_NN.py:526-529
            elif var == 'feature_layer':
                vals[:, a:b] = bv
            else:
                vals[a:b] = bv
        results.append(vals)
    return results


# ---------------------------------------------------------------- uncertainty
def binary_uncertainty_filter(posts, B):
    """PW_NNAL.binary_uncertainty_filter (PW_NNAL.py:671-681).  Tie rule of this build:
    stable (lower index first); the reference's np.argsort default is not stable."""
    return np.argsort(np.abs(np.array(posts) - 0.5), kind='stable')[:B]


def compute_entropy(pmfs):
    """NNAL_tools.compute_entropy (NNAL_tools.py:71-85): in-place +10e-8 on exact zeros."""
    pmfs[pmfs == 0] += 10e-8
    return -np.sum(pmfs * np.log(pmfs), axis=0)


def uncertainty_filtering(posteriors, B):
    """NNAL_tools.uncertainty_filtering (NNAL_tools.py:22-36): in-place +1e-8 on zeros."""
    posteriors[posteriors == 0] += 1e-8
    ent = -np.sum(posteriors * np.log(posteriors), axis=0)
    return np.argsort(-ent, kind='stable')[:B]


def bin_uncertainty_filter_multimg(expr, model, sess, all_padded_imgs, pool_inds, B,
                                   x_feed_dict={}):
    """PW_NNAL.bin_uncertainty_filter_multimg (PW_NNAL.py:684-736)."""
    s = len(pool_inds)
    sizes = [len(p) for p in pool_inds]
    m = len(all_padded_imgs[0]) - 1
    per_img = [[] for _ in range(s)]
    for i in range(s):
        if sizes[i] == 0:
            continue
        stats = [[expr.train_stats[i, 2 * j], expr.train_stats[i, 2 * j + 1]] for j in range(m)]
        per_img[i] = list(batch_eval(model, sess, all_padded_imgs[i][:-1], pool_inds[i],
                                     expr.pars['patch_shape'], expr.pars['ntb'], stats,
                                     'posteriors', None, x_feed_dict)[0])
    allp = np.concatenate(per_img)
    if len(x_feed_dict) > 0:
        return allp
    order = np.argsort(np.abs(allp - 0.5), kind='stable')[:B]
    sel = global2local_inds(order, sizes)
    return sel, [np.array(per_img[i])[sel[i]] for i in range(s)]


# ---------------------------------------------------------------- Fisher
def shrink_gradient(grad, method='sum'):
    """NNAL_tools.shrink_gradient(...,'sum') (NNAL_tools.py:784-796, ravel at :831):
    s_t = (sum(gW_t) + sum(gb_t)) / (prod(W_t.shape) + len(b_t)); the two np.sum run in the
    arrays' dtype (fp32), the division in float64."""
    assert method == 'sum'
    L = len(grad) // 2
    out = np.zeros(L)
    for t in range(L):
        gw, gb = grad[2 * t], grad[2 * t + 1]
        out[t] = (np.sum(gw) + np.sum(gb)) / (np.prod(gw.shape) + len(gb))
    return np.ravel(out)


def gen_A_matrices(expr, model, sess, sel_patches, sel_posts, diag_load=1e-5):
    """PW_NNAL.gen_A_matrices (PW_NNAL.py:738-816): per sample, three-way branch on the
    posterior, one `sess.run(grad_posts[j])` per needed class at batch 1, shrink, then
    A_i = (1-p) g0 g0^T + p g1 g1^T + diag_load I."""
    L = len(model.grad_posts['1']) // 2
    out = []
    for i in range(len(sel_posts)):
        feed = {model.x: np.expand_dims(sel_patches[i], axis=0), model.keep_prob: 1.}
        p = sel_posts[i]
        if p < 1e-6:
            p = 0.
            g0 = shrink_gradient(sess.run(model.grad_posts['0'], feed_dict=feed))
            g1 = 0.
        elif p > 1 - 1e-6:
            p = 1.
            g0 = 0.
            g1 = shrink_gradient(sess.run(model.grad_posts['1'], feed_dict=feed))
        else:
            g0 = shrink_gradient(sess.run(model.grad_posts['0'], feed_dict=feed))
            g1 = shrink_gradient(sess.run(model.grad_posts['1'], feed_dict=feed))
        Ai = (1. - p) * np.outer(g0, g0) + p * np.outer(g1, g1)
        out.append(Ai + np.eye(L) * diag_load)
    return out


def shrunk_grads(model, sess, patch):
    """(g0, g1) of one patch, both classes always (used for tolerance studies)."""
    feed = {model.x: np.expand_dims(patch, axis=0), model.keep_prob: 1.}
    return (shrink_gradient(sess.run(model.grad_posts['0'], feed_dict=feed)),
            shrink_gradient(sess.run(model.grad_posts['1'], feed_dict=feed)))


def entropy_query(expr, model, sess, padded_imgs, pool_inds):
    """CNN_query(..., 'entropy') (PW_NNAL.py:51-65)."""
    posts = batch_eval(model, sess, padded_imgs, pool_inds, expr.pars['patch_shape'],
                       expr.pars['ntb'], expr.pars['stats'], 'posteriors')[0]
    return np.argsort(np.abs(posts - .5), kind='stable')[:expr.pars['k']]


def sample_query_dstr(q_dstr, k, rng_draws):
    """NNAL_tools.sample_query_dstr, `replacement=True` branch (NNAL_tools.py:844-871) with the
    k uniform draws passed in (the reference takes them from the global np.random)."""
    q = np.array(q_dstr, dtype=float)
    q[q < 0] = 0.
    Q = np.unique(q.cumsum().searchsorted(rng_draws))
    Q[Q == len(q)] = len(q) - 1
    return Q
