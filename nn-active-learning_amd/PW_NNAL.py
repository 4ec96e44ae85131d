"""Patch-wise query strategies of the scored path (reference: PW_NNAL.py): `entropy` and `fi`
branches of CNN_query / query_multimg, the uncertainty filters and gen_A_matrices, with the
reference's signatures.  `sess` is a device.DeviceSession, `model` a device.DeviceModel."""
import numpy as np

from . import NNAL_tools, PW_NN, patch_utils


def binary_uncertainty_filter(posts, B):
    """PW_NNAL.py:671-681: the B posteriors closest to 0.5 (ties: lower index first)."""
    return np.argsort(np.abs(np.array(posts) - 0.5), kind='stable')[:B]


def device_uncertainty_filter(sess, posts, B, with_keys=False):
    """Same selection on the device (alq_score_entropy + alq_topk_uncertain) for posteriors that
    are already resident: `posts` float32 device tensor [n] -> int64 device tensor [B]
    (with_keys: also their keys |p - .5| as a float64 device tensor [B], ascending)."""
    import ctypes as C
    from ._lib import check
    torch = sess.torch
    sess.bind_stream()
    n = int(posts.numel())
    B = min(int(B), n)
    keys = sess.empty((n,), torch.float64)
    check(sess.lib.alq_score_entropy(sess.ctx, C.c_void_p(posts.data_ptr()), n, C.c_void_p(keys.data_ptr()), None))
    work = sess.empty((sess.lib.alq_topk_work_bytes(n),), torch.uint8)
    out = sess.empty((B,), torch.int64)
    check(sess.lib.alq_topk_uncertain(sess.ctx, C.c_void_p(keys.data_ptr()), n, B, C.c_void_p(out.data_ptr()),
                                      C.c_void_p(work.data_ptr())))
    if with_keys:
        return out, keys.index_select(0, out)
    return out


def bin_uncertainty_filter_multimg(expr, model, sess, all_padded_imgs, pool_inds, B, x_feed_dict={}, _vols=None):
    """PW_NNAL.py:684-736: posteriors of every subject's pool voxels (per-subject stats from
    expr.train_stats), then the B most uncertain over the concatenation, split back per subject.

    Under torch.distributed (one process per GPU, volumes replicated) every rank evaluates one contiguous block of the
    concatenated pool (pool_shard.work_block) and the posterior vector is assembled on every rank by one all-reduce of
    owner-filled entries (pool_shard.allgather_rows: x + 0.0 is exact), after which this function continues exactly as
    in one process - per-patch results do not depend on how the pool is cut into device passes."""
    from . import pool_shard
    s = len(pool_inds)
    sizes = [len(p) for p in pool_inds]
    m = len(all_padded_imgs[0]) - 1
    n = int(np.sum(sizes))
    a, b = pool_shard.work_block(n)
    allp = np.zeros(n)
    off = 0
    for i in range(s):
        lo, hi = max(a, off) - off, min(b, off + sizes[i]) - off      # this rank's part of subject i, local positions
        if hi > lo:
            stats = [[expr.train_stats[i, 2 * j], expr.train_stats[i, 2 * j + 1]] for j in range(m)]
            v = None
            if _vols is not None:          # (not a reference argument) per-subject volumes the caller keeps on the device
                if i not in _vols:
                    _vols[i] = patch_utils.DeviceVolumes(sess, all_padded_imgs[i][:-1])
                v = _vols[i]
            allp[off + lo:off + hi] = PW_NN.batch_eval(model, sess, all_padded_imgs[i][:-1], np.asarray(pool_inds[i])[lo:hi],
                                                       expr.pars['patch_shape'], expr.pars['ntb'], stats,
                                                       'posteriors', None, x_feed_dict, _vols=v, _first_sample=off + lo)[0]
        elif sizes[i] > 0:
            PW_NN.mc_dropout_args(model, x_feed_dict)       # another rank's subject: keep the RNG streams in step
        off += sizes[i]
    if (a, b) != (0, n):
        allp = pool_shard.allgather_rows(n, np.arange(a, b), allp[a:b], sess)
    if len(x_feed_dict) > 0:
        return allp
    order = binary_uncertainty_filter(allp, B)
    sel_inds = patch_utils.global2local_inds(order, sizes)
    ends = np.cumsum(sizes)
    sel_posts = [allp[ends[i] - sizes[i]:ends[i]][sel_inds[i]] for i in range(s)]
    return sel_inds, sel_posts


def gen_A_matrices(expr, model, sess, sel_patches, sel_posts, diag_load=1e-5):
    """PW_NNAL.py:738-816: conditional Fisher matrices A_i = (1-p) g0 g0^T + p g1 g1^T + diag_load I
    of the (already normalised) patches, with the reference's saturation branches on `sel_posts`.
    Returns a list of float64 [L, L] arrays.  One batched device pass replaces the per-sample
    `sess.run(model.grad_posts[j])` + `shrink_gradient` loop."""
    sel_posts = np.asarray(sel_posts, dtype=np.float64)
    n = len(sel_posts)
    if n == 0:
        return []
    x = np.asarray(sel_patches)
    res = model.fisher(x.reshape((n,) + model.in_shape), p1=sel_posts, diag_load=diag_load)
    idx = list(getattr(model, 'grad_layer_idx', range(model.L)))
    if idx == list(range(model.L)):
        return [res['A'][i] for i in range(n)]
    # `grad_layers` subset (NN.py:627-633): A_size = len(grad_posts['1'])/2 (:751), the shrunk gradients of those layers
    # only; the same host formula as the reference (:810-814) on the device's g0, g1
    p = np.where(sel_posts < 1e-6, 0., np.where(sel_posts > 1 - 1e-6, 1., sel_posts))
    g0, g1 = res['g0'][:, idx], res['g1'][:, idx]
    eye = np.eye(len(idx)) * diag_load
    return [(1. - p[i]) * np.outer(g0[i], g0[i]) + p[i] * np.outer(g1[i], g1[i]) + eye for i in range(n)]


def _refine_features(F, B):
    """-> (refined matrix, True when the conditioning loop trimmed it down to ONE feature: the image-level query then drops
    the feature term, NNAL.py:443-445)."""
    nnz_feats = np.sum(F > 0, axis=1)
    feat_inds = np.argsort(-nnz_feats)[:int(B / 2)]
    ref_F = F[feat_inds, :]
    while np.linalg.matrix_rank(ref_F) < len(feat_inds):
        feat_inds = feat_inds[:-1]
        ref_F = F[feat_inds, :]
    single = False
    while np.linalg.cond(ref_F) > 1e6:
        feat_inds = feat_inds[:-1]
        ref_F = F[feat_inds, :]
        if len(feat_inds) == 1:
            print('Only one feature is selected.')
            single = True
            break
    return ref_F, single


def refine_feature_matrix(F, B):
    """PW_NNAL.refine_feature_matrix (PW_NNAL.py:819-849): the (at most B/2) features with the most positive entries,
    trimmed from the back until the matrix has full row rank and a condition number <= 1e6."""
    return _refine_features(F, B)[0]


def _entropy_query_single(expr, model, sess, padded_imgs, pool_inds):
    posts = PW_NN.batch_eval(model, sess, padded_imgs, pool_inds, expr.pars['patch_shape'],
                             expr.pars['ntb'], expr.pars['stats'], 'posteriors')[0]
    return binary_uncertainty_filter(posts, expr.pars['k'])


def fisher_candidates(expr, model, sess, padded_imgs, pool_inds, vols=None):
    """The device part of CNN_query(...,'fi') (PW_NNAL.py:89-136): posteriors, uncertainty filter
    to B candidates, their patches (channel-index normalisation of :125-129) and A-matrices.
    Returns (sel_inds, sel_posts, A list).  `vols`: the padded volumes already on the device (uploaded once per query)."""
    B = expr.pars['B']
    if vols is None:
        vols = patch_utils.DeviceVolumes(sess, padded_imgs)
    posts = PW_NN.batch_eval(model, sess, padded_imgs, pool_inds, expr.pars['patch_shape'],
                             expr.pars['ntb'], expr.pars['stats'], 'posteriors', _vols=vols)[0]
    if B < len(pool_inds):
        sel_inds = binary_uncertainty_filter(posts, B)
    else:
        # the reference's `posts.shape[1]` raises here; the evident intent is "all"
        sel_inds = np.arange(len(pool_inds))
    sel_posts = posts[sel_inds]
    m = len(padded_imgs)
    t = vols.gather(np.asarray(pool_inds)[sel_inds], expr.pars['patch_shape'],
                    np.asarray(expr.pars['stats'], dtype=np.float64)[:m], quirk=1)
    p1_in = sess.to_device(sel_posts.astype(np.float32), sess.torch.float32)
    out = model.fisher_device(t, len(sel_inds), p1_in, 1e-5, want=('A',))
    A = out['A'].cpu().numpy()
    return sel_inds, sel_posts, [A[i] for i in range(len(sel_inds))]


def CNN_query(expr, model, sess, padded_imgs, pool_inds, tr_inds, method_name):
    """PW_NNAL.CNN_query (PW_NNAL.py:18-166), branches `entropy` and `fi`.  Returns positions into
    `pool_inds` (the caller maps them, PW_AL.py:405-408)."""
    pool_inds = np.asarray(pool_inds)
    if method_name == 'random':
        return np.random.permutation(len(pool_inds))[:expr.pars['k']]
    if method_name == 'entropy':
        return _entropy_query_single(expr, model, sess, padded_imgs, pool_inds)
    if method_name == 'MC-entropy':
        # PW_NNAL.py:66-87.  The reference passes x_feed_dict as batch_eval's NINTH positional argument, which is
        # `mask` (PW_NN.py:365-366): the keep probability never reaches the feed and every iteration evaluates the
        # same deterministic posteriors.  Mirrored: MC_iters evaluations at keep_prob = 1, running average.
        x_feed_dict = {model.keep_prob: model.dropout_rate}
        total_posts = 0
        for i in range(expr.pars['MC_iters']):
            posts = PW_NN.batch_eval(model, sess, padded_imgs, pool_inds, expr.pars['patch_shape'], expr.pars['ntb'],
                                     expr.pars['stats'], 'posteriors', x_feed_dict)[0]
            total_posts = (posts + i * total_posts) / (i + 1)
        return np.argsort(np.abs(total_posts - .5), kind='stable')[:expr.pars['k']]
    if method_name == 'fi':
        lambda_ = expr.pars['lambda_']
        vols = patch_utils.DeviceVolumes(sess, padded_imgs)            # one upload for the whole query
        sel_inds, sel_posts, A = fisher_candidates(expr, model, sess, padded_imgs, pool_inds, vols)
        ref_F = None
        if lambda_ > 0:
            # PW_NNAL.py:138-150: features of the candidates, refined to a well-conditioned full-row-rank subset and
            # centred (the SDP's equality block needs X q = 0 to hold for the uniform q).  The reference evaluates them
            # for lambda_ = 0 too and never uses them there (NNAL_tools.py:626,646); skipped.
            F = PW_NN.batch_eval(model, sess, padded_imgs, pool_inds[sel_inds], expr.pars['patch_shape'],
                                 expr.pars['ntb'], expr.pars['stats'], 'feature_layer', _vols=vols)[0]
            ref_F = refine_feature_matrix(F, expr.pars['B'])
            ref_F = ref_F - np.mean(ref_F, axis=1, keepdims=True)
        soln = NNAL_tools.SDP_query_distribution(A, lambda_, ref_F, expr.pars['k'])
        q_opt = np.array(soln['x'][:len(sel_inds)]).ravel()
        Q_inds = NNAL_tools.sample_query_dstr(q_opt, expr.pars['k'], replacement=True)
        return sel_inds[Q_inds]
    raise NotImplementedError("query method %r is outside the scored path (entropy, fi)" % (method_name,))


def _features_device(expr, model, sess, padded_mods, inds, stats):
    """feature_layer of the voxels `inds` of one subject as a device tensor [n, fdim] fp32 (memory order of the layer:
    a fixed permutation of the reference's flatten order, irrelevant to dot products and norms)."""
    torch = sess.torch
    vols = patch_utils.DeviceVolumes(sess, padded_mods)
    inds = np.asarray(inds)
    out = sess.empty((len(inds), model.feature_dim), torch.float32)
    for a in range(0, len(inds), PW_NN._CHUNK):
        b = min(len(inds), a + PW_NN._CHUNK)
        t = vols.gather(inds[a:b], expr.pars['patch_shape'], np.asarray(stats, dtype=np.float64)[:len(padded_mods)], quirk=1)
        _, _, feat = model.forward_device(t, b - a, False, True)
        out[a:b] = feat
    return out


def _row_norms(sess, F):
    import ctypes as C
    from ._lib import check
    nrm = sess.empty((int(F.shape[0]),), sess.torch.float64)
    check(sess.lib.alq_row_norms(sess.ctx, C.c_void_p(F.data_ptr()), int(F.shape[0]), int(F.shape[1]), C.c_void_p(nrm.data_ptr())))
    return nrm


def _cosine_sims(sess, A, na, Bm, nb):
    """[len(A), len(Bm)] float64 device tensor of cosine similarities (alq_cosine_sims)."""
    import ctypes as C
    from ._lib import check
    S = sess.empty((int(A.shape[0]), int(Bm.shape[0])), sess.torch.float64)
    check(sess.lib.alq_cosine_sims(sess.ctx, C.c_void_p(A.data_ptr()), int(A.shape[0]), C.c_void_p(Bm.data_ptr()), int(Bm.shape[0]),
                                   int(A.shape[1]), C.c_void_p(na.data_ptr()), C.c_void_p(nb.data_ptr()), C.c_void_p(S.data_ptr())))
    return S


def _subject_stats(expr, i, m, attr='train_stats'):
    st = getattr(expr, attr)
    return [[st[i, 2 * j], st[i, 2 * j + 1]] for j in range(m)]


def rep_entropy_query(expr, model, sess, all_padded_imgs, pool_inds):
    """query_multimg 'rep-entropy' (PW_NNAL.py:284-351): among the B most uncertain voxels, greedily pick the k whose
    feature vectors best "represent" the rest of the pool - each step adds the candidate maximising
    sum_r max_{c in Q + candidate} cos(f_r, f_c) over the remaining pool voxels r.  The similarity block is one device
    GEMM (alq_cosine_sims, fp64 accumulation like the reference's float64 NumPy), a greedy step one streaming pass that
    scores ALL candidates at once (alq_colsum_max) where the reference scores one candidate per Python iteration."""
    import ctypes as C
    from ._lib import check
    torch = sess.torch
    sess.bind_stream()
    k, B = expr.pars['k'], expr.pars['B']
    m = len(all_padded_imgs[0]) - 1
    s = len(pool_inds)
    F = [_features_device(expr, model, sess, all_padded_imgs[i][:-1], pool_inds[i], _subject_stats(expr, i, m)) if len(pool_inds[i])
         else sess.empty((0, model.feature_dim), torch.float32) for i in range(s)]
    sel_inds, sel_posts = bin_uncertainty_filter_multimg(expr, model, sess, all_padded_imgs, pool_inds, B)
    F_unc = torch.cat([F[i].index_select(0, sess.to_device(np.asarray(sel_inds[i]), torch.int64)) for i in range(s) if len(sel_inds[i]) > 0])
    rem = []
    for i in range(s):
        keep = np.setdiff1d(np.arange(len(pool_inds[i])), np.asarray(sel_inds[i], dtype=np.int64))
        rem.append(F[i].index_select(0, sess.to_device(keep, torch.int64)))
    F_rem = torch.cat(rem)
    nB, nR = int(F_unc.shape[0]), int(F_rem.shape[0])
    S = _cosine_sims(sess, F_rem, _row_norms(sess, F_rem), F_unc, _row_norms(sess, F_unc))        # [remaining, uncertain]
    work = sess.empty((sess.lib.alq_colsum_work_bytes(nR, nB),), torch.uint8)
    scores = sess.empty((nB,), torch.float64)
    cmax = sess.empty((nR,), torch.float64)
    Q, taken = [], np.zeros(nB, bool)
    for it in range(min(k, nB)):
        check(sess.lib.alq_colsum_max(sess.ctx, C.c_void_p(S.data_ptr()), nR, nB, C.c_void_p(cmax.data_ptr()) if it else None, None,
                                      C.c_void_p(scores.data_ptr()), C.c_void_p(work.data_ptr())))
        sc = scores.cpu().numpy()
        sc[taken] = -np.inf
        j = int(np.argmax(sc))                      # first maximum among the candidates left, in their original order (:339-343)
        Q.append(j)
        taken[j] = True
        check(sess.lib.alq_take_colmax(sess.ctx, C.c_void_p(S.data_ptr()), nR, nB, j, 0 if it else 1, C.c_void_p(cmax.data_ptr())))
    local = patch_utils.global2local_inds(Q, [len(sel_inds[i]) for i in range(s)])
    return [np.array(sel_inds[i])[local[i]] for i in range(s)]


def core_set_query(expr, model, sess, all_padded_imgs, pool_inds, labeled_inds):
    """query_multimg 'core-set' (PW_NNAL.py:353-451): k-centre greedy on cosine similarity - start from every pool voxel's
    largest similarity to the labelled set, then k times take the voxel LEAST similar to anything chosen or labelled.
    As in the reference the labelled side is the LAST subject of `labeled_inds` only (its loop variable `i` is read
    after the loop, :399-404), evaluated in batches of 1000 (NN.gen_batch_inds) with `expr.labeled_stats`."""
    import ctypes as C
    from . import NN
    from ._lib import check
    torch = sess.torch
    sess.bind_stream()
    k = expr.pars['k']
    m = len(all_padded_imgs[0]) - 1
    s = len(pool_inds)
    sizes = [len(p) for p in pool_inds]
    F_u = torch.cat([_features_device(expr, model, sess, all_padded_imgs[i][:-1], pool_inds[i], _subject_stats(expr, i, m))
                     if sizes[i] else sess.empty((0, model.feature_dim), torch.float32) for i in range(s)])
    n = int(F_u.shape[0])
    norms_u = _row_norms(sess, F_u)
    sims = torch.full((n,), -np.inf, dtype=torch.float64, device=sess.device)
    i = len(labeled_inds) - 1
    nT = len(labeled_inds[i])
    if expr.labeled_paths == expr.train_paths:
        lab_mods = all_padded_imgs[i][:-1]
    else:
        from . import PW_AL
        lab_mods = PW_AL.load_and_pad(list(expr.labeled_paths[i][:-1]) + [expr.labeled_paths[i][-1]], expr.pars['patch_shape'])[:-1]
    for batch in NN.gen_batch_inds(nT, 1000):
        F_T = _features_device(expr, model, sess, lab_mods, np.array(labeled_inds[i])[batch], _subject_stats(expr, i, m, 'labeled_stats'))
        St = _cosine_sims(sess, F_T, _row_norms(sess, F_T), F_u, norms_u)                           # [labelled batch, pool]
        check(sess.lib.alq_fold_rowmax(sess.ctx, C.c_void_p(St.data_ptr()), int(F_T.shape[0]), n, C.c_void_p(sims.data_ptr())))
    Q = []
    for _ in range(min(k, n)):
        q_ind = int(np.argmin(sims.cpu().numpy()))
        Q.append(q_ind)
        Sq = _cosine_sims(sess, F_u[q_ind:q_ind + 1], norms_u[q_ind:q_ind + 1], F_u, norms_u)      # [1, pool]
        check(sess.lib.alq_fold_rowmax(sess.ctx, C.c_void_p(Sq.data_ptr()), 1, n, C.c_void_p(sims.data_ptr())))
        sims[q_ind] = np.inf
    return patch_utils.global2local_inds(Q, sizes)


def query_multimg(expr, model, sess, all_padded_imgs, pool_inds, labeled_inds, method_name):
    """PW_NNAL.query_multimg (PW_NNAL.py:169-629), branches `entropy` (:226-230) and `fi`
    (:547-627).  Returns, per subject, positions into that subject's pool_inds."""
    k = expr.pars['k']
    B = expr.pars['B']
    sizes = [len(p) for p in pool_inds]
    if method_name == 'random':
        inds = np.random.permutation(int(np.sum(sizes)))[:k]
        return patch_utils.global2local_inds(inds, sizes)
    if method_name == 'entropy':
        return bin_uncertainty_filter_multimg(expr, model, sess, all_padded_imgs, pool_inds, k)[0]
    if method_name in ('MC-entropy', 'BALD'):
        # PW_NNAL.py:232-282: MC_iters dropout passes over the whole pool (keep_prob = model.dropout_rate through
        # batch_eval's x_feed_dict), running means with the reference's update (new + i * mean) / (i + 1).
        # MC-entropy ranks by |mean posterior - .5|; BALD by H(mean posterior) - mean H(posterior), zeros lifted by 1e-6.
        def bin_entropy(p):
            a, b = p.copy(), 1 - p
            a[a == 0] += 1e-6
            b[b == 0] += 1e-6
            return -a * np.log(a) - b * np.log(b)
        feed = {model.keep_prob: model.dropout_rate}
        mean_p, mean_h = 0, 0
        for it in range(expr.pars['MC_iters']):
            p = bin_uncertainty_filter_multimg(expr, model, sess, all_padded_imgs, pool_inds, k, feed)
            mean_p = (p + it * mean_p) / (it + 1)
            if method_name == 'BALD':
                mean_h = (bin_entropy(p) + it * mean_h) / (it + 1)
        if method_name == 'MC-entropy':
            order = np.argsort(np.abs(mean_p - .5), kind='stable')
        else:
            order = np.argsort(-(bin_entropy(mean_p) - mean_h), kind='stable')
        return patch_utils.global2local_inds(order[:k], sizes)
    if method_name == 'rep-entropy':
        return rep_entropy_query(expr, model, sess, all_padded_imgs, pool_inds)
    if method_name == 'core-set':
        return core_set_query(expr, model, sess, all_padded_imgs, pool_inds, labeled_inds)
    if method_name == 'fi':
        from . import pool_shard
        dvols = {}                                                       # one upload per subject for the whole query
        sel_inds, sel_posts = bin_uncertainty_filter_multimg(expr, model, sess, all_padded_imgs, pool_inds, B, _vols=dvols)
        m = len(all_padded_imgs[0]) - 1
        stats = np.asarray(expr.train_stats, dtype=np.float64)
        nsel = [len(s_) for s_ in sel_inds]
        ncand = int(np.sum(nsel))
        a, b = pool_shard.work_block(ncand)            # this rank's block of the candidate list (volumes are replicated)
        rows = []
        off = 0
        for i in range(len(pool_inds)):
            lo, hi = max(a, off) - off, min(b, off + nsel[i]) - off
            if hi > lo:
                vols = dvols.get(i) or patch_utils.DeviceVolumes(sess, all_padded_imgs[i][:m])
                t = vols.gather(np.asarray(pool_inds[i])[np.asarray(sel_inds[i])[lo:hi]], expr.pars['patch_shape'],
                                stats[i, :2 * m], quirk=0)                 # slab rule, patch_utils.py:1203-1207
                p1_in = sess.to_device(np.asarray(sel_posts[i][lo:hi], dtype=np.float32), sess.torch.float32)
                out = model.fisher_device(t, hi - lo, p1_in, 1e-3, want=('A',))   # diag_load 1e-3, :578
                rows.append(out['A'].cpu().numpy())
            off += nsel[i]
        A_rows = np.concatenate(rows) if rows else np.zeros((0, model.L, model.L))
        if (a, b) != (0, ncand):
            A_rows = pool_shard.allgather_rows(ncand, np.arange(a, b), A_rows, sess)
        A = [A_rows[j] for j in range(ncand)]
        # PW_NNAL.py:600-614: 'CVXOPT' -> SDP_query_distribution, 'MOSEK' -> solve_FIAL_SDP; both end in
        # the same A-optimal-design problem, solved here by NNAL_tools' own routine (parity unpinned)
        if expr.pars.get('SDP_solver', 'CVXOPT') == 'MOSEK':
            q_opt = np.asarray(NNAL_tools.solve_FIAL_SDP(A)[0]).ravel()
        else:
            soln = NNAL_tools.SDP_query_distribution(A, expr.pars['lambda_'], [], k)
            q_opt = np.array(soln['x'][:len(A)]).ravel()
        draws = NNAL_tools.sample_query_dstr(q_opt, k, replacement=True)
        local = patch_utils.global2local_inds(draws, [len(s) for s in sel_inds])
        return [np.array(sel_inds[i])[local[i]] for i in range(len(sel_inds))]
    raise NotImplementedError("query method %r is outside the scored path (entropy, fi)" % (method_name,))
