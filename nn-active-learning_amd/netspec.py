"""Network definitions (the reference's layer-dict schemas) and reproducible weight draws - product copy.

The networks SURVEY.md section 8 fixes for BASELINE.json's configs, written in the reference's own schemas, and
the He-normal draw in the order of `perform_assign_ops('init')` (reference `NN.py:476-504`, std rule
`NN.py:1449-1470`), so that the bench and the loop harness build their models without touching `oracle/`
(the oracle keeps its own copy; `tests/test_capi_and_host.py` checks that both draw identical weights).

* NET-A / NET-B use the `NN.CNN` schema (`NN.py:96-110`):
  ``{name: [depth, 'conv', [kh, kw]] | [depth, 'fc'] | [[window, stride], 'pool']}``; NET-B is `NN.create_PW1`
  (`NN.py:1328-1336`).
* NET-C uses the `NN_extended.CNN` schema (`NN_extended.py:103-124`): ``{name: [type, specs, op_order]}`` plus
  ``skips`` (`NN_extended.py:139-146`).
"""
from collections import OrderedDict

import numpy as np


def net_a(nclass=2):
    """3-layer 2-D CNN of configs 1-2 (SURVEY.md §8 NET-A)."""
    return OrderedDict([
        ('conv1', [16, 'conv', [5, 5]]),
        ('max1', [[2, 2], 'pool']),
        ('conv2', [32, 'conv', [3, 3]]),
        ('max2', [[2, 2], 'pool']),
        ('fc1', [nclass, 'fc']),
    ])


def net_b(nclass=2):
    """`create_PW1` layer dict, NN.py:1328-1336."""
    return OrderedDict([
        ('conv1', [24, 'conv', [5, 5]]),
        ('conv2', [32, 'conv', [5, 5]]),
        ('max1', [[2, 2], 'pool']),
        ('conv3', [48, 'conv', [3, 3]]),
        ('conv4', [96, 'conv', [3, 3]]),
        ('max2', [[2, 2], 'pool']),
        ('fc1', [4096, 'fc']),
        ('fc2', [4096, 'fc']),
        ('fc3', [nclass, 'fc']),
    ])


def net_b_small(nclass=2, width=64):
    """NET-B topology with narrow fc layers: used where the 168 MB weight set is too slow for CI."""
    d = net_b(nclass)
    d['fc1'] = [width, 'fc']
    d['fc2'] = [width, 'fc']
    return d


def net_c(nclass=2):
    """3-D U-Net-style net with fc head (SURVEY.md §8 NET-C), NN_extended schema.

    Returns (layer_dict, skips)."""
    k3 = [3, 3, 3]
    s2 = [2, 2, 2]
    layers = OrderedDict([
        ('enc1', ['conv', [8, k3], 'MA']),
        ('pool1', ['pool', s2]),
        ('enc2', ['conv', [16, k3], 'MA']),
        ('pool2', ['pool', s2]),
        ('bott', ['conv', [32, k3], 'MA']),
        ('up1', ['conv_transpose', [16, k3, s2], 'M']),
        ('dec1', ['conv', [16, k3], 'MA']),
        ('up2', ['conv_transpose', [8, k3, s2], 'M']),
        ('dec2', ['conv', [8, k3], 'MA']),
        ('fc', ['fc', [nclass]]),
    ])
    skips = [[0, [8], 'con'], [2, [6], 'con']]
    return layers, skips


def net_c_2d(nclass=2):
    """2-D analogue of NET-C (conv2d / conv2d_transpose / pool2d) for small fast tests."""
    k = [3, 3]
    s = [2, 2]
    layers = OrderedDict([
        ('enc1', ['conv', [8, k], 'MA']),
        ('pool1', ['pool', s]),
        ('enc2', ['conv', [16, k], 'MA']),
        ('up1', ['conv_transpose', [8, k, s], 'M']),
        ('dec1', ['conv', [8, k], 'MA']),
        ('fc', ['fc', [nclass]]),
    ])
    skips = [[0, [4], 'con']]
    return layers, skips


def is_extended(layer_dict):
    """The two schemas differ in the position of the type string."""
    first = next(iter(layer_dict.values()))
    return isinstance(first[0], str)


def param_shapes(layer_dict, in_shape, skips=()):
    """TF variable shapes [(name, W_shape, b_shape)] in creation order.

    `in_shape` = input shape without batch, channels last ((H,W,C) or (D,H,W,C)).
    Follows NN.py:272-277,313-318 and NN_extended.py:397-404,436-441,555-561.
    """
    ext = is_extended(layer_dict)
    names = list(layer_dict.keys())
    spatial = list(in_shape[:-1])
    ch = in_shape[-1]
    flat = None
    out = []
    src_ch = {}
    for i, name in enumerate(names):
        spec = layer_dict[name]
        ltype = spec[0] if ext else spec[1]
        for (src, dsts, kind) in skips:
            if i in dsts:
                if kind == 'con':
                    ch = ch + src_ch[src]
        if ltype == 'conv':
            cout = spec[1][0] if ext else spec[0]
            k = list(spec[1][1]) if ext else list(spec[2])
            out.append((name, tuple(k + [ch, cout]), (cout,)))
            ch = cout
        elif ltype == 'conv_transpose':
            cout, k, s = spec[1]
            out.append((name, tuple(list(k) + [cout, ch]), (cout,)))
            spatial = [a * b for a, b in zip(spatial, s)]
            ch = cout
        elif ltype == 'pool':
            stride = spec[1] if ext else [spec[0][1]] * len(spatial)
            spatial = [-(-a // b) for a, b in zip(spatial, stride)]
        elif ltype == 'fc':
            cout = spec[1][0] if ext else spec[0]
            if flat is None:
                flat = int(np.prod(spatial)) * ch
            out.append((name, (cout, flat), (cout, 1)))
            flat = cout
        else:
            raise ValueError(ltype)
        src_ch[i] = ch
    return out


def he_init(layer_dict, in_shape, seed, skips=(), bias_std=0.0, dtype=np.float32):
    """Draws weights in the order of `perform_assign_ops('init')` (NN.py:473-504).

    One global ``np.random.seed(seed)``; per parameterised layer in creation order:
    ``W = sqrt(2/n) * randn(*W_shape)`` with ``n = prod(W_shape[:-1])`` for conv
    (NN.py:481-489; NN_extended.py:1595-1599 uses the same product for 2-D/3-D and
    transpose kernels) or ``n = W_shape[1]`` for fc (NN.py:490-496); biases zero
    (NN.py:499-501).  `bias_std` > 0 additionally draws N(0, bias_std) biases AFTER all
    weights (not a reference feature: used by fixtures to exercise the bias terms).
    Returns OrderedDict name -> [W, b] in TF layouts.
    """
    rs = np.random.RandomState(seed)
    pars = OrderedDict()
    shapes = param_shapes(layer_dict, in_shape, skips)
    for name, wshape, bshape in shapes:
        if len(wshape) > 2:
            n = int(np.prod(wshape[:-1]))
        else:
            n = wshape[1]
        std = np.sqrt(2.0 / n)
        W = (std * rs.randn(*wshape)).astype(dtype)
        pars[name] = [W, np.zeros(bshape, dtype=dtype)]
    if bias_std > 0:
        for name, wshape, bshape in shapes:
            pars[name][1] = (bias_std * rs.randn(*bshape)).astype(dtype)
    return pars


def count_params(pars):
    return int(sum(W.size + b.size for W, b in pars.values()))
