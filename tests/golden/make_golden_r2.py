#!/usr/bin/env python3
"""Round-2 goldens, again produced by running the REFERENCE's own Python in the build container:

    python tests/golden/make_golden_r2.py        ->  tests/golden/r2_host.npz, tests/golden/r2_sdp.npz

Executed from /root/reference, unmodified:
  * PW_AL.get_stats, PW_AL.gen_multimg_inds              (volume statistics, grid indices: SURVEY.md 8f-4)
  * NN.gen_batch_inds                                     (fine-tune batching, 8f-2)
  * PW_NNAL.refine_feature_matrix                         (feature refinement in front of the lambda > 0 SDP, 8f-1)
  * NNAL_tools.inequality_cvx_matrix, append_zero and the matrix assembly of NNAL_tools.SDP_query_distribution up
    to the `solvers.sdp(...)` call, for lambda = 0 and lambda > 0                     (the SDP's problem STATEMENT)

Two names of absent third-party modules are bound to stand-ins so that those lines run:
  * `nrrd.read(path)` -> (array, {}) from an in-memory table (pynrrd is absent; the functions only index the arrays);
  * `cvxopt.matrix(x)` -> a NumPy array of the same 2-D shape (1-D input becomes a column, `.trans()` transposes) and
    `solvers.sdp(c, Gs, hs, A, b)` -> records its arguments and returns.  cvxopt keeps the shape of a 2-D NumPy array
    (only that reading makes the reference's G blocks (d+1)^2 x (n+d), which is what solvers.sdp requires), so the
    recorded c, G_k, h_k, A, b ARE the conic program the reference states; the cvxopt ITERATE stays unpinned."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402


class M(np.ndarray):
    def trans(self):
        return np.asarray(self).T.view(M)


def matrix(x):
    a = np.array(x, dtype=np.float64)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)
    return a.view(M)


def main():
    NNAL_tools, patch_utils, PW_NN, PW_NNAL = make_golden.import_reference()
    import NN
    import PW_AL
    out = {}

    # ------------------------------------------------------------------ volumes: stats + grid indices
    rs = np.random.RandomState(2101)
    table, paths = {}, []
    shapes = [(7, 9, 4), (8, 6, 5)]
    for s_, shp in enumerate(shapes):
        sub = []
        for j in range(2):
            name = 'sub%d_mod%d.nrrd' % (s_, j)
            table[name] = (rs.randn(*shp) * (j + 1.5) + s_).astype(np.float64)
            sub.append(name)
        mask = rs.randint(0, 2, size=shp).astype(np.float64)
        mask[rs.rand(*shp) < 0.15] = np.nan
        table['sub%d_mask.nrrd' % s_] = mask
        sub.append('sub%d_mask.nrrd' % s_)
        paths.append(sub)
    PW_AL.nrrd.read = lambda p: (table[p], {})
    for name, arr in table.items():
        out['vol_' + name] = arr
    out['stats'] = PW_AL.get_stats(paths)
    for sp in (2, 3):
        inds, labels = PW_AL.gen_multimg_inds(paths, sp)
        for i in range(len(paths)):
            out['grid%d_inds_%d' % (sp, i)] = np.array(inds[i], dtype=np.int64)
            out['grid%d_labels_%d' % (sp, i)] = np.array(labels[i], dtype=np.float64)
    # three modalities: the [i, j*m] indexing of get_stats runs out of its 2m columns (IndexError in the reference)
    paths3 = [[paths[0][0], paths[0][1], paths[0][0], paths[0][2]]]
    try:
        PW_AL.get_stats(paths3)
        out['stats_m3_raises'] = np.bool_(False)
    except IndexError:
        out['stats_m3_raises'] = np.bool_(True)
    # one modality
    out['stats_m1'] = PW_AL.get_stats([[paths[1][0], paths[1][2]]])

    # ------------------------------------------------------------------ batching
    np.random.seed(31)
    for k, (n, b) in enumerate([(23, 5), (20, 5), (3, 8)]):
        bt = NN.gen_batch_inds(n, b)
        out['batches_%d_n' % k] = np.array([n, b])
        out['batches_%d_flat' % k] = np.array([i for bb in bt for i in bb], dtype=np.int64)
        out['batches_%d_lens' % k] = np.array([len(bb) for bb in bt], dtype=np.int64)

    # ------------------------------------------------------------------ feature refinement
    F = np.maximum(rs.randn(30, 16), 0.)
    F[5] = F[7]                   # a dependent row: rank handling
    F[11] = 0.
    out['refine_F'] = F
    out['refine_out'] = PW_NNAL.refine_feature_matrix(F.copy(), 16)
    np.savez_compressed(os.path.join(HERE, 'r2_host.npz'), **out)

    # ------------------------------------------------------------------ the SDP statement
    NNAL_tools.matrix = matrix
    rec = {}

    class Solvers(object):
        options = {}

        @staticmethod
        def sdp(c, Gs=None, hs=None, A=None, b=None):
            rec.update(c=np.asarray(c), Gs=[np.asarray(g) for g in Gs], hs=[np.asarray(h) for h in hs], A=np.asarray(A), b=np.asarray(b))
            return {'status': 'recorded', 'x': None}
    NNAL_tools.solvers = Solvers
    sdp = {}
    n, L = 7, 3
    g = rs.randn(n, 2, L) * 0.3
    p = rs.rand(n)
    A = [(1 - p[i]) * np.outer(g[i, 0], g[i, 0]) + p[i] * np.outer(g[i, 1], g[i, 1]) + 1e-3 * np.eye(L) for i in range(n)]
    sdp['A'] = np.stack(A)
    X = rs.randn(2, n)
    X -= X.mean(axis=1, keepdims=True)
    sdp['X_pool'] = X
    for tag, lam in (('l0', 0.), ('l1', 0.35)):
        NNAL_tools.SDP_query_distribution(A, lam, X, 3)
        sdp[tag + '_lambda'] = np.float64(lam)
        sdp[tag + '_c'] = rec['c']
        sdp[tag + '_A'] = rec['A']
        sdp[tag + '_b'] = rec['b']
        sdp[tag + '_nG'] = np.int64(len(rec['Gs']))
        for k in range(len(rec['Gs'])):
            sdp[tag + '_G%d' % k] = rec['Gs'][k]
            sdp[tag + '_h%d' % k] = rec['hs'][k]
    sdp['append_zero'] = NNAL_tools.append_zero(A[0])
    np.savez_compressed(os.path.join(HERE, 'r2_sdp.npz'), **sdp)
    print('wrote r2_host.npz (%d arrays), r2_sdp.npz (%d arrays)' % (len(out), len(sdp)))


if __name__ == '__main__':
    main()
