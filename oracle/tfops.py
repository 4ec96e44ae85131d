"""TensorFlow-1.x op semantics restated on torch-CPU, plus loop-level numpy twins.

TEST INFRASTRUCTURE (see oracle/__init__.py).  TensorFlow is a third-party
dependency of the reference that is absent from this image (version unpinned,
API implies <= 1.15); the definitions below are the published ones:

* `tf.nn.conv2d/conv3d(..., padding='SAME')`: cross-correlation; per spatial dim
  ``out = ceil(in/s)``, ``pad_total = max((out-1)*s + k - in, 0)``,
  ``pad_before = pad_total // 2`` (the odd unit goes to the END).
  Call sites: NN.py:285-290, NN_extended.py:416-426.
* `tf.nn.max_pool/max_pool3d(..., padding='SAME')`: same geometry, padding never
  wins the max.  Call sites: NN.py:1473-1477, NN_extended.py:1665-1674.
* `tf.nn.conv2d_transpose/conv3d_transpose(x, W, output_shape, strides)` with
  padding 'SAME': the gradient w.r.t. the input of the SAME conv that maps
  `output_shape` -> `x.shape` with filter `W[k..., out, in]`, i.e.
  ``y[p] = sum_{q,t : s*q + t - pad_before = p} x[q] * W[t]``.
  Call site: NN_extended.py:574-587 (output_shape = s*in).
* flatten: `tf.transpose(x)` with no perm reverses ALL axes, then
  `tf.reshape([F, -1])` (NN.py:296-301, NN_extended.py:240-247).

All tensors here are channels-last, like the reference's placeholders.
"""
import numpy as np
import torch
import torch.nn.functional as F


def same_pads(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    lo = total // 2
    return out, lo, total - lo


def _to_cf(x):
    """channels-last [N, *sp, C] -> channels-first [N, C, *sp]."""
    nd = x.dim() - 2
    perm = [0, nd + 1] + list(range(1, nd + 1))
    return x.permute(*perm)


def _to_cl(x):
    nd = x.dim() - 2
    perm = [0] + list(range(2, nd + 2)) + [1]
    return x.permute(*perm)


def conv_same(x, W, b, strides=None):
    """x [N,*sp,Ci], W [*k,Ci,Co] (HWIO / DHWIO), b [Co]."""
    nd = x.dim() - 2
    strides = list(strides) if strides is not None else [1] * nd
    k = list(W.shape[:nd])
    pads = []
    for d in range(nd):
        _, lo, hi = same_pads(x.shape[1 + d], k[d], strides[d])
        pads.append((lo, hi))
    xc = _to_cf(x)
    # F.pad pads the LAST dim first
    flat = []
    for lo, hi in reversed(pads):
        flat += [lo, hi]
    xc = F.pad(xc, flat)
    wperm = [nd + 1, nd] + list(range(nd))
    w = W.permute(*wperm)
    fn = F.conv2d if nd == 2 else F.conv3d
    y = fn(xc, w, None, stride=strides)
    return _to_cl(y) + b


def max_pool_same(x, window, strides):
    nd = x.dim() - 2
    window = list(window)
    strides = list(strides)
    pads = []
    for d in range(nd):
        _, lo, hi = same_pads(x.shape[1 + d], window[d], strides[d])
        pads.append((lo, hi))
    xc = _to_cf(x)
    flat = []
    for lo, hi in reversed(pads):
        flat += [lo, hi]
    xc = F.pad(xc, flat, value=float('-inf'))
    fn = F.max_pool2d if nd == 2 else F.max_pool3d
    return _to_cl(fn(xc, window, strides))


def conv_transpose_same(x, W, b, strides):
    """x [N,*sp,Ci], W [*k,Co,Ci], output spatial = strides*in (NN_extended.py:574-579)."""
    nd = x.dim() - 2
    strides = list(strides)
    k = list(W.shape[:nd])
    xc = _to_cf(x)
    wperm = [nd + 1, nd] + list(range(nd))   # -> [Ci, Co, *k]
    w = W.permute(*wperm)
    fn = F.conv_transpose2d if nd == 2 else F.conv_transpose3d
    full = fn(xc, w, None, stride=strides)   # length (in-1)*s + k per dim
    sl = [slice(None), slice(None)]
    for d in range(nd):
        out = x.shape[1 + d] * strides[d]
        _, lo, _ = same_pads(out, k[d], strides[d])
        # a full result shorter than lo+out only happens for k < s (never used by the reference)
        sl.append(slice(lo, lo + out))
    y = full[tuple(sl)]
    for d in range(nd):
        want = x.shape[1 + d] * strides[d]
        if y.shape[2 + d] < want:
            padspec = [0, 0] * (nd - 1 - d) + [0, want - y.shape[2 + d]]
            y = F.pad(y, padspec)
    return _to_cl(y) + b


def flatten_tf(x):
    """[N,*sp,C] -> [F, N] with the full-axis-reversal order of `tf.transpose`."""
    nd = x.dim()
    rev = x.permute(*reversed(range(nd)))
    return rev.reshape(-1, x.shape[0])


def softmax_cols(z):
    """`tf.transpose(tf.nn.softmax(tf.transpose(z)))` for z [c, N] (NN.py:185-188)."""
    return torch.softmax(z.t(), dim=1).t()


# --------------------------------------------------------------------------
# loop-level numpy twins: small cases only, used to pin the torch calls above
# --------------------------------------------------------------------------

def naive_conv_same(x, W, b, strides=None):
    x = np.asarray(x, np.float64)
    W = np.asarray(W, np.float64)
    nd = x.ndim - 2
    strides = list(strides) if strides is not None else [1] * nd
    k = W.shape[:nd]
    geo = [same_pads(x.shape[1 + d], k[d], strides[d]) for d in range(nd)]
    out_sp = [g[0] for g in geo]
    y = np.zeros([x.shape[0]] + out_sp + [W.shape[-1]])
    for o in np.ndindex(*out_sp):
        for t in np.ndindex(*k):
            src = [o[d] * strides[d] + t[d] - geo[d][1] for d in range(nd)]
            if any(s < 0 or s >= x.shape[1 + d] for d, s in enumerate(src)):
                continue
            y[(slice(None),) + o] += x[(slice(None),) + tuple(src)] @ W[t]
    return y + np.asarray(b, np.float64).reshape(-1)


def naive_max_pool_same(x, window, strides):
    x = np.asarray(x, np.float64)
    nd = x.ndim - 2
    geo = [same_pads(x.shape[1 + d], window[d], strides[d]) for d in range(nd)]
    out_sp = [g[0] for g in geo]
    y = np.full([x.shape[0]] + out_sp + [x.shape[-1]], -np.inf)
    for o in np.ndindex(*out_sp):
        for t in np.ndindex(*window):
            src = [o[d] * strides[d] + t[d] - geo[d][1] for d in range(nd)]
            if any(s < 0 or s >= x.shape[1 + d] for d, s in enumerate(src)):
                continue
            y[(slice(None),) + o] = np.maximum(y[(slice(None),) + o],
                                              x[(slice(None),) + tuple(src)])
    return y


def naive_conv_transpose_same(x, W, b, strides):
    x = np.asarray(x, np.float64)
    W = np.asarray(W, np.float64)
    nd = x.ndim - 2
    k = W.shape[:nd]
    out_sp = [x.shape[1 + d] * strides[d] for d in range(nd)]
    lo = [same_pads(out_sp[d], k[d], strides[d])[1] for d in range(nd)]
    y = np.zeros([x.shape[0]] + out_sp + [W.shape[nd]])
    for q in np.ndindex(*x.shape[1:1 + nd]):
        for t in np.ndindex(*k):
            p = [q[d] * strides[d] + t[d] - lo[d] for d in range(nd)]
            if any(v < 0 or v >= out_sp[d] for d, v in enumerate(p)):
                continue
            # W[t] is [Co, Ci]
            y[(slice(None),) + tuple(p)] += x[(slice(None),) + q] @ W[t].T
    return y + np.asarray(b, np.float64).reshape(-1)


# ---------------------------------------------------------------- dropout
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def dropout_keep(seed, layer, sample_ids, n_elems, keep_prob):
    """Keep mask [len(sample_ids), n_elems] (bool) of the build's dropout.  `tf.nn.dropout(x, keep_prob)` (NN.py:169-171)
    is x * keep / keep_prob with keep ~ Bernoulli(keep_prob) from TensorFlow's random stream, which cannot be
    reproduced outside TensorFlow: the build defines its own counter-based mask (csrc/train.hip dropout_kernel) keyed
    (seed, layer index, sample id, element index in the layer output's memory order), and this is its restatement."""
    with np.errstate(over='ignore'):
        sid = np.asarray(sample_ids, dtype=np.uint64)[:, None]
        e = np.arange(n_elems, dtype=np.uint64)[None, :]
        key = np.uint64(seed) ^ (sid * np.uint64(0xD1B54A32D192ED03)) ^ (np.uint64(layer) << np.uint64(56))
        h = _splitmix64(_splitmix64(key) + e)
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u < np.float32(keep_prob)
