"""Oracle model: the reference's `CNN` graphs restated on torch-CPU autograd.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows `NN.CNN` (/root/reference/NN.py:147-188 layer loop, :258-340 layer
builders, :621-645 get_gradients) and `NN_extended.CNN`
(/root/reference/NN_extended.py:187-295 layer loop, :366-468 and :537-601 layer
builders, :1011-1035 get_gradients, :1119-1216 skip connections).

The object also plays the part of the reference's `model` + `sess` pair for the
golden generator: the attributes `x, keep_prob, posteriors, prediction,
feature_layer, grad_posts` are opaque handles, and `OracleSession.run(fetch,
feed_dict)` evaluates them, so the reference's own `batch_eval` /
`gen_A_matrices` can be executed verbatim against it.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import netspec, tfops


class _Dim(object):
    def __init__(self, v):
        self.value = v


class Handle(object):
    """Opaque stand-in for a TF tensor / placeholder (hashable, has .shape[i].value)."""

    def __init__(self, name, shape=()):
        self.name = name
        self.shape = [_Dim(s) for s in shape]

    def __repr__(self):
        return '<oracle handle %s>' % self.name


class OracleModel(object):

    def __init__(self, layer_dict, in_shape, pars, skips=(), feature_layer=None,
                 dtype=torch.float32, threads=None):
        self.layer_dict = layer_dict
        self.ext = netspec.is_extended(layer_dict)
        self.names = list(layer_dict.keys())
        self.in_shape = tuple(in_shape)
        self.skips = [list(s) for s in skips]
        self.dtype = dtype
        self.feature_idx = feature_layer
        if threads:
            torch.set_num_threads(threads)
        # parameters in creation order, TF layouts
        self.var_names = list(pars.keys())
        self.params = OrderedDict()
        for name, (W, b) in pars.items():
            self.params[name] = [
                torch.tensor(np.asarray(W), dtype=dtype, requires_grad=True),
                torch.tensor(np.asarray(b), dtype=dtype, requires_grad=True)]
        self.nlayers_par = len(self.params)
        self.nclass = list(self.params.values())[-1][0].shape[0]
        # reference-style handles
        self.x = Handle('x')
        self.keep_prob = Handle('keep_prob')
        self.dropout_rate = 1.
        self.posteriors = Handle('posteriors')
        self.prediction = Handle('prediction')
        fdim = self._feature_dim()
        self.feature_layer = Handle('feature_layer', (fdim,))
        self.grad_posts = {str(j): [Handle('grad_%d_%d' % (j, t))
                                    for t in range(2 * self.nlayers_par)]
                           for j in range(self.nclass)}
        self.var_dict = self.params

    # ------------------------------------------------------------------
    def _ltype(self, name):
        spec = self.layer_dict[name]
        return spec[0] if self.ext else spec[1]

    def _feature_dim(self):
        if self.feature_idx is None:
            return 0
        with torch.no_grad():
            x = torch.zeros((1,) + self.in_shape, dtype=self.dtype)
            return int(self._graph(x)['feature_layer'].shape[0])

    def _dropped(self, out, i, drop):
        """tf.nn.dropout on layer i's output (NN.py:169-171) with the build's counter-based mask (tfops.dropout_keep).
        `drop` = dict(layers, keep_prob, seed, first_sample)."""
        if drop is None or drop['keep_prob'] >= 1. or i not in drop['layers']:
            return out
        kp = drop['keep_prob']
        if out.dim() == 2:                       # fc activations are [features, N]
            F, N = out.shape
            keep = tfops.dropout_keep(drop['seed'], i, drop.get('first_sample', 0) + np.arange(N), F, kp).T
        else:                                    # [N, (D,) H, W, C]: element index = memory order of one sample
            N = out.shape[0]
            per = int(np.prod(out.shape[1:]))
            keep = tfops.dropout_keep(drop['seed'], i, drop.get('first_sample', 0) + np.arange(N), per, kp).reshape(tuple(out.shape))
        k = torch.as_tensor(np.ascontiguousarray(keep)).to(out.dtype)
        return out * k * torch.tensor(np.float32(1.0) / np.float32(kp), dtype=torch.float32).to(out.dtype)

    def _graph(self, x, drop=None):
        """x: [N, *in_shape] tensor -> dict of graph nodes."""
        names = self.names
        out = x
        flat = False
        feats = None
        sources = {}
        src_idx = [s[0] for s in self.skips]
        for i, name in enumerate(names):
            spec = self.layer_dict[name]
            ltype = self._ltype(name)
            last = (i == len(names) - 1)
            nxt = None if last else self._ltype(names[i + 1])
            # skip connections: NN_extended.py:1119-1190 (sizes always agree here)
            for (src, dsts, kind) in self.skips:
                if i in dsts:
                    if kind == 'sum':
                        out = out + sources[src]
                    else:
                        # concat_outputs(curr, prev) puts the EARLIER output first (:1207-1214)
                        out = torch.cat((sources[src], out), dim=out.dim() - 1)
            if self.ext:
                order = spec[2] if len(spec) > 2 else 'M'
                for op in order:
                    if op == 'M':
                        out = self._main_op(name, ltype, spec[1], out)
                    elif op == 'A':
                        out = torch.relu(out)
                    else:
                        raise NotImplementedError('op %r (batch-norm) is outside the scored path' % op)
            else:
                if ltype == 'conv':
                    W, b = self.params[name]
                    out = torch.relu(tfops.conv_same(out, W, b))          # NN.py:285-290
                elif ltype == 'pool':
                    w, s = spec[0]
                    nd = out.dim() - 2
                    out = tfops.max_pool_same(out, [w] * nd, [s] * nd)     # NN.py:1473-1477
                elif ltype == 'fc':
                    W, b = self.params[name]
                    out = W @ out + b                                       # NN.py:322-324
                    if not last:
                        out = torch.relu(out)                               # NN.py:326-327
                else:
                    raise ValueError(ltype)
                # NN.CNN flattens INSIDE add_conv/add_pool, i.e. before the feature marker
                if ltype in ('conv', 'pool') and nxt == 'fc':
                    out = tfops.flatten_tf(out)
                    flat = True
            # NN.py:169-171 drops the layer's output as the next layer (and the flatten inside add_conv) sees it; the device
            # keys the mask by the element's position in the UN-flattened activation, so drop before flattening
            if drop is not None and i in drop['layers'] and drop['keep_prob'] < 1.:
                if (not self.ext) and ltype in ('conv', 'pool') and nxt == 'fc':
                    raise NotImplementedError('dropout on the conv/pool layer in front of the first fc (NN.CNN schema)')
                out = self._dropped(out, i, drop)
            if i in src_idx:
                sources[i] = out
            if self.feature_idx is not None and i == self.feature_idx:
                feats = out
            if self.ext and (not last) and ltype in ('conv', 'pool') and nxt == 'fc':
                out = tfops.flatten_tf(out)                                 # NN_extended.py:237-247
                flat = True
        assert out.dim() == 2, 'the scored path needs an fc head (get_gradients: NN_extended.py:1025)'
        post = tfops.softmax_cols(out)
        return {'output': out, 'posteriors': post, 'feature_layer': feats}

    def _main_op(self, name, ltype, specs, out):
        if ltype == 'conv':
            W, b = self.params[name]
            strides = specs[2] if len(specs) > 2 else None
            return tfops.conv_same(out, W, b, strides)                      # NN_extended.py:416-426
        if ltype == 'conv_transpose':
            W, b = self.params[name]
            return tfops.conv_transpose_same(out, W, b, specs[2])           # NN_extended.py:574-587
        if ltype == 'pool':
            return tfops.max_pool_same(out, specs, specs)                   # NN_extended.py:453-468
        if ltype == 'fc':
            W, b = self.params[name]
            return W @ out + b                                              # NN_extended.py:449-451
        raise ValueError(ltype)

    # ------------------------------------------------------------------
    def _as_input(self, x):
        x = torch.as_tensor(np.asarray(x)).to(self.dtype)   # placeholder is tf.float32
        return x.reshape((-1,) + self.in_shape)

    def forward(self, x, drop=None):
        with torch.no_grad():
            g = self._graph(self._as_input(x), drop)
        res = {'output': g['output'].numpy(),
               'posteriors': g['posteriors'].numpy(),
               'prediction': g['posteriors'].argmax(dim=0).numpy()}      # NN.py:618-619
        if g['feature_layer'] is not None:
            res['feature_layer'] = g['feature_layer'].numpy()
        return res

    def loss_and_grads(self, x, y_onehot, drop=None):
        """Mean softmax cross-entropy (NN.py:583-588: tf.reduce_mean(softmax_cross_entropy_with_logits)) of a batch
        and its gradient w.r.t. every variable [W1, b1, ...] (what the optimizer's minimize() differentiates)."""
        g = self._graph(self._as_input(x), drop)
        z = g['output']                                          # [c, N]
        y = torch.as_tensor(np.asarray(y_onehot)).to(z.dtype)
        logp = z - torch.logsumexp(z, dim=0, keepdim=True)
        loss = -(y * logp).sum(dim=0).mean()
        plist = [p for pair in self.params.values() for p in pair]
        grads = torch.autograd.grad(loss, plist, allow_unused=True)
        return float(loss.detach()), [np.zeros(p.shape, dtype=p.detach().numpy().dtype) if gr is None else gr.numpy()
                             for p, gr in zip(plist, grads)]

    def grad_log_post(self, j, x, drop=None):
        """`tf.gradients(tf.log(posteriors[j, 0]), trainable_variables)` (NN.py:639-645).

        Returns the 2L arrays [W1, b1, ..., WL, bL] in TF variable shapes."""
        g = self._graph(self._as_input(x), drop)
        score = torch.log(g['posteriors'][j, 0])
        plist = [p for pair in self.params.values() for p in pair]
        grads = torch.autograd.grad(score, plist, allow_unused=True)
        out = []
        for p, gr in zip(plist, grads):
            out.append(np.zeros(p.shape, dtype=p.detach().numpy().dtype) if gr is None
                       else gr.numpy())
        return out


class OracleSession(object):
    """`sess.run(fetch, feed_dict=...)` over an OracleModel's handles."""

    def __init__(self, model):
        self.model = model
        self.calls = 0

    def run(self, fetch, feed_dict=None):
        m = self.model
        x = feed_dict[m.x]
        if m.keep_prob in feed_dict:
            assert float(feed_dict[m.keep_prob]) == 1., 'only keep_prob = 1 is on the scored path'
        self.calls += 1
        if isinstance(fetch, list):
            for j, handles in m.grad_posts.items():
                if fetch is handles:
                    return m.grad_log_post(int(j), x)
            raise KeyError('unknown fetch list')
        res = m.forward(x)
        return res[fetch.name]
