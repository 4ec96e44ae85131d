"""Torch restatement of the FACTORED algorithm the HIP kernels implement (test helper).

The reference ships every per-sample weight gradient to the host and sums it there
(PW_NNAL.py:773-807 + NNAL_tools.py:784-796).  The kernels never materialise those gradients;
they use, per parameterised layer t with input a and pre-activation cotangent d:

  conv            sum(dW)+sum(db) = sum_x dsum[x] * (box_k(asum)[x] + 1)
  conv_transpose  sum(dW)+sum(db) = sum_q asum[q] * sum_t dsum[s*q + t - lo] + sum_p dsum[p]
  fc              sum(dW)+sum(db) = (sum_i d_i) * (sum_j a_j + 1)

with asum/dsum the channel sums, and ONE backward pass with the unit cotangent (+1,-1) on the
two logits, because d log p_j / dz = e_j - p makes the class-0 / class-1 cotangents
p1*(+1,-1) and p0*(-1,+1).  This file states that algorithm with torch ops so that the
identity can be checked against the oracle in fp64 on the CPU, and so that GPU intermediates
can be compared layer by layer.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import tfops


def _box_same(field, k):
    """zero-padded k-window sum of a [N,*sp] field, SAME geometry, stride 1."""
    nd = field.dim() - 1
    x = field.unsqueeze(1)
    pads = []
    for d in reversed(range(nd)):
        _, lo, hi = tfops.same_pads(field.shape[1 + d], k[d], 1)
        pads += [lo, hi]
    x = F.pad(x, pads)
    w = torch.ones((1, 1) + tuple(k), dtype=field.dtype)
    fn = F.conv2d if nd == 2 else F.conv3d
    return fn(x, w)[:, 0]


def _strided_box(dsum, k, s, in_sp):
    """B[q] = sum_t dsum[s*q + t - lo] over in-range positions, q on the INPUT grid."""
    nd = dsum.dim() - 1
    x = dsum.unsqueeze(1)
    pads = []
    for d in reversed(range(nd)):
        _, lo, hi = tfops.same_pads(dsum.shape[1 + d], k[d], s[d])
        pads += [lo, hi]
    x = F.pad(x, pads)
    w = torch.ones((1, 1) + tuple(k), dtype=dsum.dtype)
    fn = F.conv2d if nd == 2 else F.conv3d
    y = fn(x, w, stride=tuple(s))[:, 0]
    assert tuple(y.shape[1:]) == tuple(in_sp)
    return y


def factored_unit_scores(model, x, details=None, flips=None):
    """Returns (p [2,N], S [N,L], sizes [L]): S_t = sum of ALL entries of d(z0-z1)/d(theta_t).
    If `details` is a dict it receives per-layer tensors: 'out' (list over ALL layers, output after
    activation, channels-last), and per parameterised layer 'delta' (masked cotangent), 'asum',
    'dsum', 'pre' (pre-activation), 'relu' (bool).
    `flips`: {layer name: bool array shaped like that layer's pre-activation}: ReLU decisions to INVERT (the unit passes
    although its input is <= 0, or is cut although > 0) - what an implementation does whose rounding puts a near-zero
    ReLU input on the other side (relu_flip_explains below).

    `model` is an oracle.model.OracleModel (its graph is re-walked here with hooks on every
    parameterised layer's input and pre-activation output)."""
    xt = model._as_input(x)
    N = xt.shape[0]
    recs = []

    # re-implementation of the walk with explicit pre-activation capture
    names = model.names
    out = xt
    layer_outs = []
    pool_inputs = {}
    sources = {}
    src_idx = [s[0] for s in model.skips]
    for i, name in enumerate(names):
        spec = model.layer_dict[name]
        ltype = model._ltype(name)
        last = (i == len(names) - 1)
        nxt = None if last else model._ltype(names[i + 1])
        for (src, dsts, kind) in model.skips:
            if i in dsts:
                out = out + sources[src] if kind == 'sum' else torch.cat((sources[src], out), out.dim() - 1)
        if ltype in ('conv', 'conv_transpose', 'fc'):
            W, b = [p.detach() for p in model.params[name]]
            a_in = out
            if ltype == 'conv':
                strides = None
                if model.ext and len(spec[1]) > 2:
                    strides = spec[1][2]
                pre = tfops.conv_same(out, W, b, strides)
                relu = ('A' in spec[2]) if model.ext else True
            elif ltype == 'conv_transpose':
                pre = tfops.conv_transpose_same(out, W, b, spec[1][2])
                relu = 'A' in spec[2]
            else:
                pre = W @ out + b
                relu = (('A' in spec[2]) if len(spec) > 2 else False) if model.ext else (not last)
            pre.requires_grad_(True)
            pre.retain_grad()
            recs.append(dict(name=name, type=ltype, a=a_in.detach(), pre=pre, W=W, b=b, spec=spec, relu=relu))
            if relu and flips is not None and name in flips:
                keep = (pre.detach() > 0) ^ torch.as_tensor(np.asarray(flips[name], dtype=bool))
                # a unit inverted to 'passes' outputs |pre| (derivative 1): the engine that let it pass computed a positive value, and a
                # max-pool behind the layer compares it with the zeros of its window (csrc/ref64.hip does the same)
                out = pre * keep.to(pre.dtype) + ((pre.detach().abs() - pre.detach()) * keep.to(pre.dtype))
            else:
                out = torch.relu(pre) if relu else pre
        elif ltype == 'pool':
            if flips is not None and ('pool:' + name) in flips:
                # a near-tie of a max-pool window decided the other way: the runner-up is lifted past the winner by less than
                # the implementations' rounding noise (relu_flip_explains), which moves the value by nothing but re-routes the
                # whole backward path of that window to another voxel
                out = out + torch.as_tensor(np.asarray(flips['pool:' + name]), dtype=out.dtype)
            pool_inputs[name] = out.detach()
            if model.ext:
                out = tfops.max_pool_same(out, spec[1], spec[1])
            else:
                nd = out.dim() - 2
                out = tfops.max_pool_same(out, [spec[0][0]] * nd, [spec[0][1]] * nd)
        if i in src_idx:
            sources[i] = out
        layer_outs.append(out.detach())
        if (not last) and ltype in ('conv', 'pool') and nxt == 'fc':
            out = tfops.flatten_tf(out)
    z = out                                     # [2, N]
    p = tfops.softmax_cols(z.detach())
    (z[0] - z[1]).sum().backward()
    S = torch.zeros((N, len(recs)), dtype=xt.dtype)
    sizes = []
    for t, r in enumerate(recs):
        d = r['pre'].grad
        W, b = r['W'], r['b']
        sizes.append(int(W.numel() + b.shape[0]))
        if r['type'] == 'fc':
            S[:, t] = d.sum(0) * (r['a'].sum(0) + 1.)
        elif r['type'] == 'conv':
            nd = d.dim() - 2
            k = list(W.shape[:nd])
            dsum = d.sum(-1)
            asum = r['a'].sum(-1)
            S[:, t] = (dsum * (_box_same(asum, k) + 1.)).reshape(N, -1).sum(1)
        else:
            nd = d.dim() - 2
            k = list(W.shape[:nd])
            s = list(r['spec'][1][2])
            dsum = d.sum(-1)
            asum = r['a'].sum(-1)
            B = _strided_box(dsum, k, s, asum.shape[1:])
            S[:, t] = (asum * B).reshape(N, -1).sum(1) + dsum.reshape(N, -1).sum(1)
    if details is not None:
        details['out'] = [o.numpy() for o in layer_outs]
        details['delta'] = [r['pre'].grad.numpy() for r in recs]
        details['asum'] = [(r['a'].sum(0) if r['type'] == 'fc' else r['a'].sum(-1)).numpy() for r in recs]
        details['dsum'] = [(r['pre'].grad.sum(0) if r['type'] == 'fc' else r['pre'].grad.sum(-1)).numpy()
                           for r in recs]
        details['types'] = [r['type'] for r in recs]
        details['names'] = [r['name'] for r in recs]
        details['pre'] = [r['pre'].detach().numpy() for r in recs]
        details['relu'] = [bool(r['relu']) for r in recs]
        details['pool_in'] = {k: v.numpy() for k, v in pool_inputs.items()}
        details['pool_k'] = {n: (model.layer_dict[n][1] if model.ext else [model.layer_dict[n][0][0]] * (pool_inputs[n].dim() - 2))
                             for n in pool_inputs}
    return p.numpy(), S.numpy(), np.array(sizes)


def fisher_from_unit(p1, S, sizes, diag_load):
    """g0, g1, A exactly as gen_A_matrices would form them from the unit-cotangent sums."""
    p1 = np.asarray(p1, np.float64)
    g = np.asarray(S, np.float64) / sizes[None, :]
    p0 = 1. - p1
    g0 = p1[:, None] * g
    g1 = -p0[:, None] * g
    N, L = g.shape
    A = np.zeros((N, L, L))
    for i in range(N):
        p = p1[i]
        a0, a1 = g0[i], g1[i]
        if p < 1e-6:
            p, a1 = 0., np.zeros(L)
        elif p > 1 - 1e-6:
            p, a0 = 1., np.zeros(L)
        A[i] = (1. - p) * np.outer(a0, a0) + p * np.outer(a1, a1) + np.eye(L) * diag_load
    return g0, g1, A


def relu_flip_explains(model64, x1, targets, diag_load, atol=2e-6, rtol=2e-5, eps=4e-6, max_units=10, max_flips=3):
    """fp64 arbiter for a disagreement between fp32-level implementations on ONE patch.

    `targets`: list of (g0 [L], g1 [L]) score vectors of that patch (e.g. two device engines).  Each must lie within
    atol + rtol |value| of the fp64 evaluation of the network (`model64`: an fp64 OracleModel), OR of an fp64 evaluation in
    which some of the patch's FRAGILE decisions are inverted.  Fragile = a ReLU input with |pre-activation| <= eps x the layer's
    rms pre-activation, i.e. one that fp32 rounding can legitimately put on either side of zero - or (round 4) a max-pool window
    (window = stride, even extents) whose two largest inputs lie within eps x the layer's rms of each other with a positive
    maximum: which of them is the arg-max decides where the window's whole cotangent goes.  At most `max_units` most fragile
    units are considered, at most `max_flips` of them inverted together.  eps = 4e-6 = 5.4 x the largest key the device arbiter
    (nnal_amd/ref64.py, same rule) measured over the decisions that explain the engines' disagreements on the bench patches.
    Returns a list, per target, of the tuple of inverted units ((layer name, flat index), ...; a pool window as
    ('pool:<name>', flat index of the lifted input)) - () = the plain fp64 value - or None when no such evaluation matches (the
    disagreement is NOT such a flip)."""
    import itertools
    x1 = np.asarray(x1)[None] if np.asarray(x1).ndim == len(model64.in_shape) else np.asarray(x1)
    det = {}
    p, S, sizes = factored_unit_scores(model64, x1, det)

    def scores(P, SS):
        g0, g1, _ = fisher_from_unit(P[1], SS, sizes, diag_load)
        return g0[0], g1[0]

    def close(t, ref):
        return all(np.all(np.abs(np.asarray(a) - b) <= atol + rtol * np.abs(b)) for a, b in zip(t, ref))

    base = scores(p, S)
    found = [() if close(t, base) else None for t in targets]
    if all(f is not None for f in found):
        return found
    cand = []
    for name, pre, relu in zip(det['names'], det['pre'], det['relu']):
        if not relu:
            continue
        rms = float(np.sqrt(np.mean(pre ** 2))) or 1.0
        flat = np.abs(pre.reshape(-1)) / rms
        for i in np.nonzero(flat <= eps)[0]:
            cand.append((flat[i], name, int(i), pre.shape))
    for name, xin in det.get('pool_in', {}).items():
        k = list(det['pool_k'][name])
        sp = xin.shape[1:-1]
        if xin.shape[0] != 1 or any(d % kk for d, kk in zip(sp, k)):
            continue                       # (ragged SAME windows are not needed by the nets this arbiter serves)
        C = xin.shape[-1]
        rms = float(np.sqrt(np.mean(xin ** 2))) or 1.0
        idx = np.arange(xin.size).reshape(xin.shape)
        shp2, perm = [1], [0]
        for d, kk in zip(sp, k):
            shp2 += [d // kk, kk]
        shp2 += [C]
        nd = len(sp)
        win_axes = [2 + 2 * a for a in range(nd)]
        cell_axes = [1 + 2 * a for a in range(nd)]
        order = [0] + cell_axes + [1 + 2 * nd] + win_axes
        v = xin.reshape(shp2).transpose(order).reshape(-1, int(np.prod(k)))
        ii = idx.reshape(shp2).transpose(order).reshape(-1, int(np.prod(k)))
        srt = np.argsort(-v, axis=1)
        top, sec = v[np.arange(len(v)), srt[:, 0]], v[np.arange(len(v)), srt[:, 1]]
        gap = (top - sec) / rms
        for w in np.nonzero((gap <= eps) & (top > 0))[0]:
            cand.append((gap[w], 'pool:' + name, int(ii[w, srt[w, 1]]), xin.shape, float(2.0 * (top[w] - sec[w]) + 1e-12 * rms)))
    cand.sort(key=lambda c: c[0])
    cand = cand[:max_units]
    for r in range(1, min(max_flips, len(cand)) + 1):
        for combo in itertools.combinations(cand, r):
            fl, shapes = {}, {}
            for c in combo:
                name, i, shp = c[1], c[2], c[3]
                if name.startswith('pool:'):
                    fl.setdefault(name, np.zeros(int(np.prod(shp))))[i] = c[4]
                else:
                    fl.setdefault(name, np.zeros(int(np.prod(shp)), bool))[i] = True
                shapes[name] = shp
            fl = {k: v.reshape(shapes[k]) for k, v in fl.items()}
            pf, Sf, _ = factored_unit_scores(model64, x1, None, flips=fl)
            ref = scores(pf, Sf)
            for j, t in enumerate(targets):
                if found[j] is None and close(t, ref):
                    found[j] = tuple((c[1], c[2]) for c in combo)
            if all(f is not None for f in found):
                return found
    return found
