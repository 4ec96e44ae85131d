"""Host-side helpers of the scoring path (reference: NNAL_tools.py).  These are NumPy in the
reference too; the device twins are alq_score_entropy / alq_fisher."""
import warnings

import numpy as np


def uncertainty_filtering(posteriors, B):
    """NNAL_tools.py:22-36.  posteriors [c, n]; guards exact zeros IN PLACE (+1e-8) like the
    reference, returns the B highest-entropy columns (ties: lower index first)."""
    posteriors[posteriors == 0] += 1e-8
    H = -np.sum(posteriors * np.log(posteriors), axis=0)
    return np.argsort(-H, kind='stable')[:B]


def compute_entropy(PMFs):
    """NNAL_tools.py:71-85.  PMFs [c, n]; +10e-8 on exact zeros, in place."""
    PMFs[PMFs == 0] += 10e-8
    return -np.sum(PMFs * np.log(PMFs), axis=0)


def idxBatch_posteriors(model, inds, expr, session, col, extra_feed_dict={}):
    """NNAL_tools.idxBatch_posteriors (NNAL_tools.py:382-448): posteriors [c, n] (float64) of the images `inds` of
    expr.imgs_path_file, evaluated in random batches of expr.pars['batch_size'] (one NN.gen_batch_inds draw from the global
    NumPy stream when n >= batch_size; the reference's n < batch_size branch iterates over bare ints and fails in
    load_winds - here it is one batch of everything and no draw)."""
    from . import NN
    batch_size = expr.pars['batch_size']
    inds = np.asarray(inds)
    n = len(inds)
    posteriors = np.zeros((model.nclass, n))
    batches = [np.arange(n).tolist()] if n < batch_size else NN.gen_batch_inds(n, batch_size)
    for inner in batches:
        X, _ = NN.load_winds(inds[inner], expr.imgs_path_file, expr.pars['target_shape'], expr.pars['mean'])
        feed = {model.x: X}
        feed.update(extra_feed_dict)
        post = session.run(model.posteriors, feed_dict=feed)
        posteriors[:, inner] = post if col else post.T
    return posteriors


def shrink_gradient(grad, method, args=None):
    """NNAL_tools.py:778-831, 'sum' and 'max' methods: one scalar per parameterised layer from a
    list [gW1, gb1, ..., gWL, gbL].  On the device path this list is never built (the kernels
    reduce per layer directly); the function is kept for callers that hold full gradients."""
    L = len(grad) // 2
    out = np.zeros(L)
    for t in range(L):
        gW, gb = np.asarray(grad[2 * t]), np.asarray(grad[2 * t + 1])
        if method == 'sum':
            # fp32 sums, then a float64 quotient: the divisor is a NumPy int64 as in the reference
            # (np.prod(shape) + len(b)); a Python int would keep the quotient in fp32 under NumPy 2
            out[t] = (np.sum(gW) + np.sum(gb)) / (np.prod(gW.shape) + len(gb))
        elif method == 'max':
            # max(..., key=abs) per array, then the plain max of the two winners (NNAL_tools.py:805-810)
            wmax = gW.ravel()[np.argmax(np.abs(gW.ravel()))]
            bmax = gb.ravel()[np.argmax(np.abs(gb.ravel()))]
            out[t] = max(wmax, bmax)
        else:
            raise NotImplementedError("shrink method %r is not on the scored path" % (method,))
    return np.ravel(out)


def append_zero(A):
    """NNAL_tools.py:833-842."""
    d = A.shape[0]
    out = np.zeros((d + 1, d + 1), dtype=A.dtype)
    out[:d, :d] = A
    return out


def sample_query_dstr(q_dstr, k, replacement=True):
    """NNAL_tools.py:844-896: draws from the query distribution with the global np.random;
    clips negatives in place (warns below -0.01)."""
    if q_dstr.min() < -.01:
        warnings.warn('Optimal q has significant negative values..')
    q_dstr[q_dstr < 0] = 0.
    if replacement:
        Q_inds = np.unique(q_dstr.cumsum().searchsorted(np.random.sample(k)))
        Q_inds[Q_inds == len(q_dstr)] = len(q_dstr) - 1
        return Q_inds
    remaining = np.arange(len(q_dstr))
    picked = []
    q = q_dstr
    while len(picked) < k:
        j = int(q.cumsum().searchsorted(np.random.sample(1))[0])
        picked.append(remaining[j])
        remaining = np.delete(remaining, j)
        q = np.delete(q, j)
        if np.all(q == 0):
            q[:] = 1.
        q = q / np.sum(q)
    return np.array(picked)


def _aopt_multiplicative(A, tol, max_iter):
    """q_i <- q_i * sqrt(d_i / tr M^-1): the classical multiplicative algorithm for the A-criterion (monotone;
    linear convergence, slow once the weights outside the support have to decay).  Kept as the independent
    cross-check of the Newton solver below."""
    n = A.shape[0]
    q = np.full(n, 1.0 / n)
    status, gap = 'unknown', np.inf
    for it in range(max_iter):
        M = np.tensordot(q, A, axes=(0, 0))
        Minv = np.linalg.inv(M)
        obj = np.trace(Minv)
        d = np.tensordot(A, Minv @ Minv, axes=([1, 2], [0, 1]))       # tr(M^-1 A_i M^-1)
        gap = d.max() / obj - 1.0
        if gap < tol:
            status = 'optimal'
            break
        q = q * np.sqrt(d / obj)
        q /= q.sum()
    return q, status, it + 1


def _svec_basis(L):
    """Orthonormal basis of the symmetric L x L matrices as columns of P [L*L, L(L+1)/2]: <X, Y> = (P^T vec X)(P^T vec Y)."""
    m = L * (L + 1) // 2
    P = np.zeros((L * L, m))
    c = 0
    for i in range(L):
        for j in range(i, L):
            if i == j:
                P[i * L + i, c] = 1.0
            else:
                P[i * L + j, c] = P[j * L + i, c] = np.sqrt(0.5)
            c += 1
    return P


def _aopt_newton(A, tol, max_iter):
    """Log-barrier Newton method for  min tr(M(q)^-1),  M(q) = sum q_i A_i,  q in the simplex - the job the
    reference gives to an interior-point SDP solver, with the structure used instead of a generic cone program:
    the Hessian of the objective is  V K V^T  with V = [svec A_i] (n x L(L+1)/2) and
    K = M^-1 (x) M^-2 + M^-2 (x) M^-1 restricted to the symmetric matrices, the barrier adds a diagonal, so a
    Newton step is one (L(L+1)/2)^2 solve through the Woodbury identity: O(n L^4) per step, ~40 steps,
    instead of thousands of first-order sweeps.  Duality gap of a centred point = n * mu."""
    from scipy.linalg import cho_factor, cho_solve
    n, L = A.shape[0], A.shape[1]
    P = _svec_basis(L)
    m = P.shape[1]
    V = A.reshape(n, L * L) @ P

    def at(q):
        M = (P @ (q @ V)).reshape(L, L)
        Mi = np.linalg.inv(M)
        return Mi, float(np.trace(Mi))

    q = np.full(n, 1.0 / n)
    Mi, obj = at(q)
    mu = obj / n
    status, steps = 'unknown', 0
    ones = np.ones(n)
    while steps < max_iter:
        steps += 1
        Mi2 = Mi @ Mi
        d = V @ (P.T @ Mi2.reshape(-1))                        # d_i = <M^-2, A_i> = -df/dq_i
        if d.max() <= obj * (1.0 + tol):                       # optimality condition d_i <= tr M^-1 for every i
            status = 'optimal'
            break
        g = -d - mu / q
        K = P.T @ (np.kron(Mi, Mi2) + np.kron(Mi2, Mi)) @ P
        K = 0.5 * (K + K.T)
        R = np.linalg.cholesky(K)
        U = V @ R
        Dinv = q * q / mu
        UD = U * Dinv[:, None]
        S = cho_factor(np.eye(m) + U.T @ UD)

        def Hinv(x):
            y = Dinv * x
            return y - UD @ cho_solve(S, U.T @ y)
        # Newton step on {sum q = 1}: H dq + nu 1 = -g.  g is -(tr M^-1 + n mu) 1 plus a residual that vanishes on
        # the central path; solving for the residual keeps the step free of the cancellation between the two
        r = g + (obj + n * mu)
        a, b = Hinv(r), Hinv(ones)
        dq = -a + (a.sum() / b.sum()) * b
        dec = float(-(r @ dq))                                  # Newton decrement squared
        neg = dq < 0
        alpha = min(1.0, 0.99 * float(np.min(-q[neg] / dq[neg]))) if neg.any() else 1.0
        phi0 = obj - mu * np.log(q).sum()
        slack = 1e-12 * abs(phi0)                               # rounding noise of phi: a decrease below it cannot be tested
        while True:
            qn = q + alpha * dq
            Mn, on = at(qn)
            if on - mu * np.log(qn).sum() <= phi0 - 0.25 * alpha * dec + slack or alpha < 1e-12:
                break
            alpha *= 0.5
        q, Mi, obj = qn / qn.sum(), Mn, on
        if dec <= 0.05 * mu * n:                                # centred for this mu: d_i <= tr M^-1 + ~n mu
            mu = max(0.2 * mu, 0.25 * tol * obj / n)
    return q, status, steps


def _aopt_newton_eq(A, c, E, tol, max_iter):
    """Log-barrier Newton method for  min tr(M(q)^-1) + c^T q  s.t.  E q = E q0 (q0 uniform),  q >= 0 - the
    feature-regularised form of the query-distribution SDP (NNAL_tools.py:626-645: c = -lambda ||x_i||^2,
    E = [X_pool; 1^T], right-hand side (0, 1)).  Same structure as `_aopt_newton` (Hessian = diagonal + V K V^T,
    Woodbury), plus the equality block through its Schur complement  E H^-1 E^T  ((d+1) x (d+1)).
    Returns (q, nu, status, steps): nu = multipliers of the equality rows."""
    from scipy.linalg import cho_factor, cho_solve
    n, L = A.shape[0], A.shape[1]
    P = _svec_basis(L)
    m = P.shape[1]
    V = A.reshape(n, L * L) @ P
    # independent equality rows only (X_pool is full row rank after refine_feature_matrix; be safe)
    Q, R = np.linalg.qr(E.T)
    keep = np.abs(np.diag(R)) > 1e-10 * max(1.0, np.abs(np.diag(R)).max())
    Eo = Q[:, keep].T                                           # orthonormal rows spanning the same constraints

    def at(q):
        M = (P @ (q @ V)).reshape(L, L)
        Mi = np.linalg.inv(M)
        return Mi, float(np.trace(Mi))

    q = np.full(n, 1.0 / n)
    f0 = Eo @ q
    Mi, tr = at(q)
    obj = tr + float(c @ q)
    mu = max(tr, abs(obj)) / n
    status, steps = 'unknown', 0
    nu = np.zeros(Eo.shape[0])
    while steps < max_iter:
        steps += 1
        Mi2 = Mi @ Mi
        d = V @ (P.T @ Mi2.reshape(-1))                        # <M^-2, A_i>
        g = -d + c - mu / q
        K = P.T @ (np.kron(Mi, Mi2) + np.kron(Mi2, Mi)) @ P
        K = 0.5 * (K + K.T)
        Rk = np.linalg.cholesky(K)
        U = V @ Rk
        Dinv = q * q / mu
        UD = U * Dinv[:, None]
        S = cho_factor(np.eye(m) + U.T @ UD)

        def Hinv(x):
            y = Dinv[:, None] * x if x.ndim == 2 else Dinv * x
            return y - UD @ cho_solve(S, U.T @ y)
        HiE = Hinv(Eo.T)                                        # n x r
        G = Eo @ HiE
        G = 0.5 * (G + G.T)
        Hig = Hinv(g)
        nu = np.linalg.solve(G, -(Eo @ Hig))
        dq = -(Hig + HiE @ nu)
        dq -= Eo.T @ (Eo @ dq)                                  # exactly tangent: rounding in the Schur solve must not leak out of E q = f
        dec = float(-(g @ dq))                                  # Newton decrement squared (E dq = 0)
        neg = dq < 0
        alpha = min(1.0, 0.99 * float(np.min(-q[neg] / dq[neg]))) if neg.any() else 1.0
        phi0 = tr + float(c @ q) - mu * np.log(q).sum()
        slack = 1e-12 * abs(phi0)
        while True:
            qn = q + alpha * dq
            Mn, trn = at(qn)
            if trn + float(c @ qn) - mu * np.log(qn).sum() <= phi0 - 0.25 * alpha * dec + slack or alpha < 1e-12:
                break
            alpha *= 0.5
        qn = qn - Eo.T @ (Eo @ qn - f0)                         # stay on the affine set to rounding
        if qn.min() > 0:
            Mn, trn = at(qn)
        else:
            qn = q + alpha * dq
        q, Mi, tr = qn, Mn, trn
        if dec <= 0.05 * mu * n:                                # centred: duality gap ~ n mu
            if n * mu <= tol * max(abs(tr + float(c @ q)), 1e-300):
                status = 'optimal'
                break
            mu *= 0.2
    # Multipliers of the caller's rows E from stationarity at the returned point, weighted to the support (where the slack
    # s_i = -d_i + c_i + (E^T nu)_i must vanish): min_nu sum_i q_i^2 s_i^2.  (The nu of the last Newton system differs from
    # it by H dq, which is not negligible against a duality gap of n mu.)
    Mi2 = Mi @ Mi
    d = V @ (P.T @ Mi2.reshape(-1))
    Ew = E * q[None, :]
    nu_E = np.linalg.lstsq(Ew.T, q * (d - c), rcond=None)[0]
    return q, nu_E, status, steps


def SDP_query_distribution(A, lambda_, X_pool, k, tol=1e-7, max_iter=20000, method='newton'):
    """Query distribution of Fisher-information AL (reference: NNAL_tools.py:612-659, with
    `inequality_cvx_matrix` :661-720 and the CVXPY twin `solve_FIAL_SDP` :576-610).

    The reference states, for A-matrices A_i (L x L, positive definite through the diagonal load), x = (q, t):

        min  sum_j t_j - lambda * sum_i q_i ||x_i||^2
        s.t. [[sum_i q_i A_i, e_j], [e_j^T, t_j]] >= 0  (j = 1..L),   diag(q) >= 0,
             sum_i q_i = 1,   and for lambda > 0 also  X_pool q = 0          (NNAL_tools.py:626-645)

    (X_pool [d, n]: the refined, zero-meaned features of the candidates, PW_NNAL.py:146-155) and hands it to cvxopt's
    interior-point SDP solver.  By the Schur complement t_j >= e_j^T M(q)^-1 e_j, so the problem is
    min_q tr(M(q)^-1) - lambda sum_i q_i ||x_i||^2 over that polytope: an A-optimal design with a linear reward for
    long feature vectors.  cvxopt, cvxpy and MOSEK are absent from this image, so this function solves THAT problem
    itself with a log-barrier Newton method on its structure (`_aopt_newton`, `_aopt_newton_eq`;
    `method='multiplicative'`: the classical first-order algorithm, lambda = 0 only).  The problem STATEMENT is
    pinned: tests/test_capi_and_host.py checks the returned (q, t) and the dual certificate built from it against
    the c, G_k, h_k, A, b that the reference's own code assembles (tests/golden/r2_sdp.npz).  The ITERATE of the
    reference's solver (tolerance, where on the optimal face it stops) is PARITY UNPINNED.

    Returns a dict like cvxopt's: 'x' = concat(q [n], t [L]), 'status', 'primal objective', 'gap' (largest violation
    of the optimality condition, relative to the objective's scale), 'y' (multipliers of the equality rows, cvxopt's
    sign: c + sum_k G_k^T z_k + A^T y = 0), 'iterations'."""
    A = np.asarray(A, dtype=np.float64)
    n, L = A.shape[0], A.shape[1]
    lam = float(lambda_) if lambda_ else 0.0
    if lam > 0:
        X = np.asarray(X_pool, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != n:
            raise ValueError('X_pool must be [d, %d] (features x candidates), got %r' % (n, X.shape))
        if method != 'newton':
            raise ValueError('the feature-regularised form is solved by the Newton method only')
        if np.abs(X.sum(axis=1)).max() > 1e-8 * max(1.0, np.abs(X).max()) * n:
            raise ValueError('X_pool rows are not zero-mean: the uniform distribution violates X_pool q = 0 '
                             '(PW_NNAL.py:148-150 centres the features before the call)')
        c = -lam * np.sum(X ** 2, axis=0)
        E = np.concatenate((X, np.ones((1, n))), axis=0)
        q, nu, status, its = _aopt_newton_eq(A, c, E, tol, min(max_iter, 500))
        name = 'log-barrier Newton, feature-regularised A-optimal design'
    else:
        c = np.zeros(n)
        nu = None
        if method == 'newton':
            q, status, its = _aopt_newton(A, tol, min(max_iter, 500))
            name = 'log-barrier Newton A-optimal design'
        elif method == 'multiplicative':
            q, status, its = _aopt_multiplicative(A, tol, max_iter)
            name = 'multiplicative A-optimal design'
        else:
            raise ValueError('unknown method %r' % (method,))
    M = np.tensordot(q, A, axes=(0, 0))
    Minv = np.linalg.inv(M)
    t = np.diag(Minv).copy()
    d = np.tensordot(A, Minv @ Minv, axes=([1, 2], [0, 1]))
    if lam > 0:
        # stationarity: -d_i + c_i + (E^T nu)_i = s_i >= 0, s_i q_i = 0
        s = -d + c + E.T @ nu
        gap = max(float(-s.min()), float(np.abs(s * q).sum())) / max(t.sum(), 1e-300)
        y = nu
    else:
        gap = d.max() / t.sum() - 1.0
        y = np.array([float(d @ q)])                            # multiplier of sum q = 1: s_i = y - d_i
    return {'x': np.concatenate((q, t)), 'status': '%s (%s; not cvxopt)' % (status, name),
            'primal objective': float(t.sum() + c @ q), 'gap': float(gap), 'iterations': its, 'y': np.asarray(y)}


def solve_FIAL_SDP(A):
    """NNAL_tools.solve_FIAL_SDP (NNAL_tools.py:576-610): same problem through the same solver here;
    returns (q, objective) as PW_NNAL.query_multimg unpacks them (PW_NNAL.py:611-614)."""
    soln = SDP_query_distribution(A, 0., [], None)
    n = len(A)
    return soln['x'][:n], soln['primal objective']
