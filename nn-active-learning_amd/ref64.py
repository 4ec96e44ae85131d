"""fp64 accuracy reference on the device (csrc/ref64.hip, alq_ref64_scores) - an accuracy TOOL, never on the scoring path.

What it answers: the scores g0, g1 of PW_NNAL.gen_A_matrices (PW_NNAL.py:757-814) are not continuous in
rounding noise - a ReLU input (or a max-pool near-tie) within rounding of a decision boundary switches a whole backward
path - so two fp32-level engines differ by up to ~1e-3 on a few per cent of the 32^3 patches.  For a set of patches this
module states, for ANY engine's scores:
  * how many patches differ from the fp64 evaluation by more than 2e-6 / 1e-4, and the largest difference;
  * for each differing patch, the FEWEST fragile decisions (|pre-activation| <= eps x the layer's rms; pool windows whose two
    largest inputs lie within eps x rms) that have to be inverted in the fp64 evaluation to reproduce the engine's scores to
    2e-6 + 2e-5 relative - 0, 1, 2, 3 - or "unexplained" when no such set among the `max_units` most fragile ones does;
  * the fragility (|pre| / rms) of the decisions that were inverted: the measured width of the window an fp32-level engine
    needs, from which `eps` is justified rather than assumed.
bench.py puts these numbers for the shipped engines AND for the exact-fp32 MFMA engine into its `accuracy` object; the GPU
tests assert that the shipped split needs no more than 1.25 x the exact-fp32 engine's count (+ a small-count allowance).
"""
import ctypes as C
import itertools

import numpy as np

from ._lib import check
from .device import ALQ_CONV, ALQ_CONVT, ALQ_FC, ALQ_POOL


class FlipT(C.Structure):
    _fields_ = [('layer', C.c_int32), ('pad', C.c_int32), ('idx', C.c_int64), ('delta', C.c_double)]


# Width of the fragility window (|pre-activation| / rms of the layer's pre-activations, pool: gap / rms of its input).  MEASURED,
# not assumed: over the first 2047 bench patches the decisions whose inversion reproduces an engine's scores (shipped 16-bit splits
# and the exact-fp32 engine alike) have keys up to 1.17e-6 (211 decisions, profiles/r06_accuracy_vs_fp64.json; bench.py reports the
# figure of every run as accuracy.vs_fp64.fragility.max_key) - a few fp32 roundings (2^-24 = 6e-8) of a 432-term pre-activation.
# eps = 4e-6 is 3.4 x that maximum (round 5 used 2e-5, 17 x): wide enough that no explanation is missed, narrow enough that an
# engine noticeably worse than fp32 would be left UNEXPLAINED; the tests assert max_key <= eps / 2 on their 512 patches.
DEFAULT_EPS = 4e-6

FLIP_DTYPE = np.dtype([('layer', np.int32), ('pad', np.int32), ('idx', np.int64), ('delta', np.float64)])
assert FLIP_DTYPE.itemsize == C.sizeof(FlipT) == 24


class Ref64(object):
    """fp64 twin of a DeviceModel's weights on the device + the evaluation calls."""

    def __init__(self, model, max_samples=128):
        self.model = model
        self.sess = model.sess
        self.lib = model.lib
        torch = self.sess.torch
        self.max_samples = int(max_samples)
        self.L = model.L
        self._arr, self._nl, self._cd = model._create_args
        nd = len(model.in_shape) - 1
        dims = [1] * (3 - nd) + list(model.in_shape[:-1]) + [model.in_shape[-1]]
        # shape walk: the spatial shape in front of every fc layer (for the flatten permutation)
        sp, ch = list(dims[:3]), dims[3]
        chans, flat = {}, None
        self._W, self._b = [], []
        t = 0
        self.sizes = []
        for i, d in enumerate(model.layers):
            if d['skip_src'] >= 0:
                ch += chans[d['skip_src']]
            if d['type'] in (ALQ_CONV, ALQ_CONVT, ALQ_FC):
                W, b = model.var_dict[model.var_names[t]]
                W = np.asarray(W, np.float64)
                b = np.asarray(b, np.float64).reshape(-1)
                self.sizes.append(W.size + b.size)
                if d['type'] == ALQ_FC:
                    if flat is None:
                        # reference flatten order: full axis reversal of [D, H, W, C] (NN.py:296-301): j = ((c W + w) H + h) D + d
                        out = W.shape[0]
                        W = W.reshape(out, ch, sp[2], sp[1], sp[0]).transpose(0, 4, 3, 2, 1).reshape(out, -1)
                    flat = d['cout']
                elif d['type'] == ALQ_CONVT:
                    sp = [a * s for a, s in zip(sp, d['s'])]
                    ch = d['cout']
                else:
                    ch = d['cout']
                self._W.append(self.sess.to_device(np.ascontiguousarray(W).reshape(-1), torch.float64))
                self._b.append(self.sess.to_device(np.ascontiguousarray(b), torch.float64))
                t += 1
            elif d['type'] == ALQ_POOL:
                sp = [-(-a // s) for a, s in zip(sp, d['s'])]
            chans[i] = ch
        self.sizes = np.asarray(self.sizes, np.float64)
        self._pW = (C.c_void_p * self.L)(*[w.data_ptr() for w in self._W])
        self._pb = (C.c_void_p * self.L)(*[b.data_ptr() for b in self._b])
        self.relu_layers = [i for i, d in enumerate(model.layers) if d['type'] != ALQ_POOL and d['relu']]
        self.pool_layers = [i for i, d in enumerate(model.layers) if d['type'] == ALQ_POOL]

    # -- raw evaluation ----------------------------------------------------------------------------------------------------
    def evaluate(self, pool, rows, flips=None, eps=0., cand_cap=0):
        """pool: device fp32 [*, elems per patch]; rows: int array of pool rows (one evaluation each, repeats allowed);
        flips: None or structured array [len(rows), F] of FLIP_DTYPE (layer < 0 = unused).  Returns dict: logits [n, 2],
        S [n, L], rms [n_layers, n], and with cand_cap > 0 'cand' = list per sample of (key, layer, pad, idx, delta) sorted by key
        plus 'cand_overflow' (samples whose list was truncated)."""
        torch = self.sess.torch
        self.sess.bind_stream()
        rows = np.asarray(rows, np.int64).reshape(-1)
        n_all = len(rows)
        out = dict(logits=np.zeros((n_all, 2)), S=np.zeros((n_all, self.L)), rms=np.zeros((self._nl, n_all)))
        cands, overflow = [], 0
        F = 0 if flips is None else int(flips.shape[1])
        for a in range(0, n_all, self.max_samples):
            b = min(n_all, a + self.max_samples)
            n = b - a
            r = self.sess.to_device(rows[a:b], torch.int64)
            lg = self.sess.empty((n, 2), torch.float64)
            S = self.sess.empty((n, self.L), torch.float64)
            rms = self.sess.empty((self._nl, n), torch.float64)
            fl = None
            if F:
                fl = self.sess.to_device(np.ascontiguousarray(flips[a:b]).view(np.uint8).reshape(-1), torch.uint8)
            cd = ck = cc = None
            if cand_cap > 0:
                cd = self.sess.empty((n * cand_cap * 24,), torch.uint8)
                ck = self.sess.empty((n, cand_cap), torch.float64)
                cc = self.sess.empty((n,), torch.int32)
            check(self.lib.alq_ref64_scores(
                self.sess.ctx, self._arr, self._nl, self._cd, self._pW, self._pb, C.c_void_p(pool.data_ptr()), C.c_void_p(r.data_ptr()), n,
                C.c_void_p(fl.data_ptr()) if fl is not None else None, F, float(eps), int(cand_cap),
                C.c_void_p(lg.data_ptr()), C.c_void_p(S.data_ptr()), C.c_void_p(rms.data_ptr()),
                C.c_void_p(cd.data_ptr()) if cd is not None else None, C.c_void_p(ck.data_ptr()) if ck is not None else None,
                C.c_void_p(cc.data_ptr()) if cc is not None else None))
            out['logits'][a:b] = lg.cpu().numpy()
            out['S'][a:b] = S.cpu().numpy()
            out['rms'][:, a:b] = rms.cpu().numpy()
            if cand_cap > 0:
                cnt = cc.cpu().numpy()
                keys = ck.cpu().numpy()
                recs = cd.cpu().numpy().view(FLIP_DTYPE).reshape(n, cand_cap)
                for i in range(n):
                    m = min(int(cnt[i]), cand_cap)
                    overflow += int(cnt[i] > cand_cap)
                    order = np.lexsort((recs['idx'][i, :m], recs['layer'][i, :m], keys[i, :m]))      # deterministic whatever the atomics did
                    cands.append([(float(keys[i, j]), int(recs['layer'][i, j]), int(recs['pad'][i, j]), int(recs['idx'][i, j]),
                                   float(recs['delta'][i, j])) for j in order])
        if cand_cap > 0:
            out['cand'] = cands
            out['cand_overflow'] = overflow
        return out

    def scores(self, logits, S, diag_load=None):
        """p1, g0, g1 (and A with diag_load) as gen_A_matrices forms them (PW_NNAL.py:770-814), in fp64."""
        z = np.asarray(logits, np.float64)
        m = z.max(axis=1, keepdims=True)
        e = np.exp(z - m)
        p1 = e[:, 1] / e.sum(axis=1)
        g = np.asarray(S, np.float64) / self.sizes[None, :]
        g0 = p1[:, None] * g
        g1 = -(1. - p1)[:, None] * g
        lo, hi = p1 < 1e-6, p1 > 1. - 1e-6
        g1[lo] = 0.
        g0[hi] = 0.
        out = dict(p1=p1, g0=g0, g1=g1)
        if diag_load is not None:
            p = np.where(lo, 0., np.where(hi, 1., p1))
            out['A'] = ((1. - p)[:, None, None] * g0[:, :, None] * g0[:, None, :] + p[:, None, None] * g1[:, :, None] * g1[:, None, :]
                        + np.eye(self.L)[None] * diag_load)
        return out

    # -- the arbiter -------------------------------------------------------------------------------------------------------
    @staticmethod
    def close(t, ref, atol=2e-6, rtol=2e-5):
        return bool(np.all(np.abs(np.asarray(t) - ref) <= atol + rtol * np.abs(ref)))

    def explain(self, pool, rows, targets, eps, max_units=10, max_flips=3, atol=2e-6, rtol=2e-5, cand_cap=256):
        """For every row r (a pool row) and every engine e: targets[e] = (g0 [n, L], g1 [n, L]) of those rows.  Returns
        (base, found, fragility): base = fp64 scores dict of the rows; found[e][i] = () when engine e agrees with the plain fp64
        value on row i, a tuple of (key, layer, pad, idx) decisions whose inversion reproduces its scores (fewest first), or None
        (unexplained: no set of <= max_flips among the max_units most fragile decisions within eps does);
        fragility = list of the keys (|pre| / rms resp. gap / rms) of every decision used in an explanation."""
        rows = np.asarray(rows, np.int64).reshape(-1)
        n = len(rows)
        ev = self.evaluate(pool, rows, eps=eps, cand_cap=cand_cap)
        base = self.scores(ev['logits'], ev['S'])
        base['cand_overflow'] = ev['cand_overflow']
        ne = len(targets)
        found = [[None] * n for _ in range(ne)]
        for e, (t0, t1) in enumerate(targets):
            for i in range(n):
                if self.close(t0[i], base['g0'][i], atol, rtol) and self.close(t1[i], base['g1'][i], atol, rtol):
                    found[e][i] = ()
        fragility = []
        for r in range(1, max_flips + 1):
            jobs = []      # (row ordinal, combo)
            for i in range(n):
                if all(found[e][i] is not None for e in range(ne)):
                    continue
                cand = ev['cand'][i][:max_units]
                for combo in itertools.combinations(cand, r):
                    jobs.append((i, combo))
            if not jobs:
                break
            fl = np.zeros((len(jobs), max_flips), FLIP_DTYPE)
            fl['layer'] = -1
            for j, (i, combo) in enumerate(jobs):
                for k, c in enumerate(combo):
                    fl[j, k] = (c[1], c[2], c[3], c[4])
            res = self.evaluate(pool, rows[[i for i, _ in jobs]], flips=fl)
            sc = self.scores(res['logits'], res['S'])
            for j, (i, combo) in enumerate(jobs):
                for e, (t0, t1) in enumerate(targets):
                    if found[e][i] is None and self.close(t0[i], sc['g0'][j], atol, rtol) and self.close(t1[i], sc['g1'][j], atol, rtol):
                        found[e][i] = tuple((c[0], c[1], c[2], c[3]) for c in combo)
                        fragility.extend(c[0] for c in combo)
        return base, found, fragility

    def engine_report(self, pool, rows, engines, eps, **kw):
        """engines: {name: (g0, g1)} of the rows.  One summary dict per engine (counts against fp64, flips needed), plus the
        measured fragility of the inverted decisions."""
        names = list(engines)
        base, found, frag = self.explain(pool, rows, [engines[k] for k in names], eps, **kw)
        rep = {}
        for e, k in enumerate(names):
            g0, g1 = engines[k]
            d = np.maximum(np.abs(g0 - base['g0']), np.abs(g1 - base['g1'])).max(axis=1)
            hist = {'0': 0, '1': 0, '2': 0, '3': 0, 'unexplained': 0}
            for f in found[e]:
                hist['unexplained' if f is None else str(len(f))] += 1
            rep[k] = {'patches': int(len(d)), 'over_2e-6': int((d > 2e-6).sum()), 'over_1e-4': int((d > 1e-4).sum()),
                      'max_abs_dg': float(d.max()), 'flips_needed': hist,
                      'unexplained_rows': [int(rows[i]) for i, f in enumerate(found[e]) if f is None]}
        rep['_fragility'] = {'decisions_inverted': len(frag), 'max_key': float(max(frag)) if frag else 0.0,
                             'eps': float(eps), 'candidate_lists_truncated': int(base['cand_overflow'])}
        return rep, base, found
