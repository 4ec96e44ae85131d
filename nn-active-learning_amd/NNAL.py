"""Image-level query strategies (reference: NNAL.py): `CNN_query` with the branches `random`, `entropy` and the
multi-class Fisher-information query `fi` (NNAL.py:312-464) - the image-level twin of PW_NNAL's binary `fi` and the
last strategy of SURVEY.md 8f-3 that reuses the scoring kernels.

What runs where: posteriors, features and the per-class gradients of the log-posteriors run on the device
(DeviceModel.forward_device, alq_param_grads per class, alq_shrink_sum for `shrink_gradient(..., 'sum')`, alq_fisher_classes
for the sum of outer products); the reference's per-sample class selection (posteriors below 1e-6 dropped and the rest
renormalised; the ten largest when ten or more remain), the feature refinement, the SDP and the draws are host NumPy
like the reference's.  RNG draws (NN.gen_batch_inds in idxBatch_posteriors and extract_features, then
sample_query_dstr) happen in the reference's order, so a seeded run picks the same queries."""
import numpy as np

from . import NN, NNAL_tools, PW_NNAL


def class_weights(x_posterior):
    """The reference's per-sample class selection (NNAL.py:363-394) on one posterior column: classes below 1e-6 are
    dropped and the rest renormalised; with ten or more left only the ten largest are kept and renormalised again.
    Returns (weights [c] = 1 / new_posts on the kept classes, 0 elsewhere; number of kept classes)."""
    p = np.array(x_posterior, dtype=np.float64)
    p[p < 1e-6] = 0.
    nz = np.where(p > 0.)[0]
    nz_posts = p[nz] / np.sum(p[nz])
    if len(nz) < 10:
        sel, new_posts = nz, nz_posts
    else:
        top = np.argsort(-nz_posts)[:10]
        sel = nz[top]
        new_posts = nz_posts[top]
        new_posts = new_posts / np.sum(new_posts)
    w = np.zeros(len(p))
    w[sel] = 1. / new_posts
    return w, len(sel)


def fi_A_matrices(model, session, sel_X, sel_posteriors):
    """The A-matrix loop of the `fi` branch (NNAL.py:336-413): A_i = sum_{j kept} g_ij g_ij^T / p'_ij + |kept| 1e-5 I with
    g_ij = shrink_gradient(d log posteriors[j] / d theta, 'sum') of sample i.  sel_posteriors [c, B] float64 (modified
    in place like the reference's view: entries below 1e-6 become 0).  Returns the list of B float64 [L', L'] arrays."""
    c, B = sel_posteriors.shape
    W = np.zeros((B, c))
    diag = np.zeros(B)
    for i in range(B):
        x_posterior = sel_posteriors[:, i]
        x_posterior[x_posterior < 1e-6] = 0.
        W[i], kept = class_weights(x_posterior)
        diag[i] = kept * 1e-5
    A = model.fisher_classes(sel_X, W, diag)
    return [A[i] for i in range(B)]


def CNN_query(model, expr, pool_inds, method_name, session, col=True, extra_feed_dict={}):
    """NNAL.CNN_query (NNAL.py:188-525), branches `random` (:297-299), `entropy` (:301-313) and `fi` (:315-464).
    Returns positions into `pool_inds`."""
    k = expr.pars['k']
    B = expr.pars['B']
    lambda_ = expr.pars['lambda_']
    pool_inds = np.asarray(pool_inds)
    if method_name == 'random':
        return np.random.permutation(len(pool_inds))[:k]
    if method_name == 'entropy':
        posteriors = NNAL_tools.idxBatch_posteriors(model, pool_inds, expr, session, col, extra_feed_dict)
        return np.argsort(-NNAL_tools.compute_entropy(posteriors), kind='stable')[:k]
    if method_name == 'fi':
        posteriors = NNAL_tools.idxBatch_posteriors(model, pool_inds, expr, session, col, extra_feed_dict)
        if B < posteriors.shape[1]:
            sel_inds = NNAL_tools.uncertainty_filtering(posteriors, B)
            sel_posteriors = posteriors[:, sel_inds]
        else:
            B = posteriors.shape[1]
            sel_posteriors = posteriors
            sel_inds = np.arange(B)
        sel_X, _ = NN.load_winds(pool_inds[sel_inds], expr.imgs_path_file, expr.pars['target_shape'], expr.pars['mean'])
        A = fi_A_matrices(model, session, sel_X, sel_posteriors)
        # features of the candidates, refined to a full-row-rank, well-conditioned subset and centred (:415-452)
        F = model.extract_features(pool_inds[sel_inds], expr, session)
        F_sel, single = PW_NNAL._refine_features(F, B)
        if single:
            lambda_ = 0                                                          # :443-445
        F_sel = F_sel - np.mean(F_sel, axis=1, keepdims=True)
        soln = NNAL_tools.SDP_query_distribution(A, lambda_, F_sel, k)
        q_opt = np.array(soln['x'][:B]).ravel()
        Q_inds = NNAL_tools.sample_query_dstr(q_opt, k, replacement=True)
        return sel_inds[Q_inds]
    raise NotImplementedError("query method %r: the image-level path has random, entropy and fi" % (method_name,))
