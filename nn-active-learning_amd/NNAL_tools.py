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


def SDP_query_distribution(A, lambda_, X_pool, k):
    """NNAL_tools.SDP_query_distribution (NNAL_tools.py:612-659) hands the A-matrices to cvxopt's
    SDP solver.  cvxopt / cvxpy / MOSEK are absent from this image and their arithmetic is a
    third-party dependency outside the scored path ("next" row), so this
    entry point states that instead of silently substituting another optimiser."""
    raise NotImplementedError(
        'the SDP query distribution needs cvxopt (NNAL_tools.py:657), which is not installed; the '
        'device path ends at the A-matrices (PW_NNAL.fisher_candidates / gen_A_matrices)')
