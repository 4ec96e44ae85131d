"""Device context ("session") and device model behind the reference's `sess` / `model` pair.

The reference hands every query function a `tf.Session` and a `CNN` object and reaches the
device through `sess.run(getattr(model, var), feed_dict)` (PW_NN.py:466,522).  Here `sess` is a
`DeviceSession` (one HIP stream on one MI355X, a libalq context) and `model` a `DeviceModel`
(a libalq model: weights + activation workspace resident in HBM).  PyTorch-ROCm is used only
for device memory, the stream and (in pool_shard.py) torch.distributed.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import ALQ_CONV, ALQ_CONVT, ALQ_FC, ALQ_POOL, LayerT, check


def _torch():
    import torch
    return torch


class Handle(object):
    """Stand-in for a TF tensor attribute of the reference model (`model.x`, `.posteriors`, ...)."""

    class _Dim(object):
        def __init__(self, v):
            self.value = v

    def __init__(self, name, shape=()):
        self.name = name
        self.shape = [Handle._Dim(s) for s in shape]

    def __repr__(self):
        return '<device handle %s>' % self.name


_default_session = None
_live = None   # weak set of sessions / models, closed at interpreter exit BEFORE the HIP runtime unloads


def _track(obj):
    global _live
    if _live is None:
        import atexit
        import weakref
        _live = weakref.WeakSet()

        def _close_all():
            objs = list(_live)
            for o in objs:                      # models first: they hold device memory of a context
                if isinstance(o, DeviceModel):
                    o.close()
            for o in objs:
                if isinstance(o, DeviceSession):
                    o.close()
        atexit.register(_close_all)
    _live.add(obj)


def default_session():
    """The process-wide session used by functions whose reference signature carries no `sess`
    (patch_utils.get_patches).  One process drives one GPU (LOCAL_RANK picks it)."""
    global _default_session
    if _default_session is None:
        import os
        _default_session = DeviceSession(int(os.environ.get('LOCAL_RANK', '0')))
    return _default_session


class DeviceSession(object):
    """One GPU, one stream.  `run(fetch, feed_dict)` keeps unported strategies working."""

    def __init__(self, device=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.AlqError('no GPU visible: the query-scoring path has no CPU fallback')
        self.torch = torch
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        self.lib = _lib.lib()
        self._ctx = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.alq_ctx_create(device, C.c_void_p(stream), C.byref(self._ctx)))
        self._stream = stream
        self.comm_world = 0          # > 0 once pool_shard.attach_comm gave this context an RCCL communicator
        _track(self)

    @property
    def ctx(self):
        return self._ctx

    def bind_stream(self):
        """Points the library at torch's CURRENT stream on this device.  Every device call interleaves libalq
        launches with torch ops and caching-allocator frees, which are ordered on torch's current stream; a
        caller inside `torch.cuda.stream(s)` would otherwise have the library race with torch's reuse of the
        buffers it was handed.  One pointer compare per call when nothing changed."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            check(self.lib.alq_ctx_set_stream(self._ctx, C.c_void_p(s)))
            self._stream = s

    def uncertainty_filter(self, posts, B, with_keys=False):
        """The B positions of `posts` (device fp32 [n]) closest to 0.5, ascending |p - .5|, ties -> lower position
        (alq_score_entropy + alq_topk_uncertain); int64 device tensor [min(B, n)] (+ their fp64 keys on request)."""
        from .PW_NNAL import device_uncertainty_filter
        return device_uncertainty_filter(self, posts, B, with_keys)

    def topk_smallest(self, keys, B):
        """Positions of the B smallest entries of a float64 device vector, ascending, ties -> lower position (bit-pattern
        order: +inf and NaN come last)."""
        torch = self.torch
        self.bind_stream()
        n = int(keys.numel())
        work = self.empty((self.lib.alq_topk_work_bytes(n),), torch.uint8)
        out = self.empty((int(B),), torch.int64)
        check(self.lib.alq_topk_uncertain(self._ctx, C.c_void_p(keys.data_ptr()), n, int(B), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(work.data_ptr())))
        return out

    # -- RCCL communicator of the sharded pool (pool_shard.attach_comm) --------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        check(self.lib.alq_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, rank, world):
        self.bind_stream()
        check(self.lib.alq_comm_init(self._ctx, C.c_char_p(uid), int(rank), int(world)))
        self.comm_world = int(world)

    def comm_destroy(self):
        check(self.lib.alq_comm_destroy(self._ctx))
        self.comm_world = 0

    def allreduce_sum_(self, t):
        """In-place all-reduce(sum) of a float64 device tensor over the context's RCCL communicator."""
        assert t.dtype == self.torch.float64 and t.is_contiguous()
        self.bind_stream()
        check(self.lib.alq_allreduce_sum(self._ctx, C.c_void_p(t.data_ptr()), t.numel()))
        return t

    def synchronize(self):
        check(self.lib.alq_ctx_synchronize(self._ctx))

    def close(self):
        if self._ctx:
            self.lib.alq_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- torch helpers --------------------------------------------------------------------
    def to_device(self, arr, dtype):
        torch = self.torch
        t = torch.as_tensor(np.ascontiguousarray(arr))
        return t.to(device=self.device, dtype=dtype)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    # -- sess.run compatibility -----------------------------------------------------------
    def run(self, fetch, feed_dict=None):
        """`sess.run(fetch, feed_dict={model.x: batch, model.keep_prob: p, model.y_: labels})` for the fetches the
        reference's query / fine-tune code uses: `model.posteriors`, `.prediction`, `.feature_layer` (PW_NN.py:522),
        `model.grad_posts[str(j)]` (PW_NNAL.py:773-807: list of 2L' gradient arrays of log posteriors[j, 0]) and
        `model.train_step` (PW_AL.py:1075-1080, :1140-1146)."""
        model = getattr(fetch, 'model', None)
        if isinstance(fetch, list):
            model = getattr(fetch[0], 'model', None) if fetch else None
            if model is None:
                raise KeyError('unknown fetch list')
            x = feed_dict[model.x]
            kp = float(feed_dict.get(model.keep_prob, 1.))
            j = fetch[0].cls
            return model.grad_log_post(x, j, keep_prob=kp)
        if model is None:
            raise KeyError('unknown fetch %r' % (fetch,))
        x = feed_dict[model.x]
        kp = float(feed_dict.get(model.keep_prob, 1.))
        if fetch.name == 'train_step':
            return model.train_on_batch(x, feed_dict[model.y_], keep_prob=kp)
        res = model.forward(x, want=(fetch.name,), keep_prob=kp)
        return res[fetch.name]

    # -- measurement hooks ----------------------------------------------------------------
    def prof_enable(self, on=True):
        """True / 1: time every launch; k > 1: the launches of every k-th Fisher pass; False / 0: off."""
        check(self.lib.alq_prof_enable(self._ctx, int(on)))

    def prof_reset(self):
        check(self.lib.alq_prof_reset(self._ctx))

    def prof_read(self):
        out = OrderedDict()
        for c in range(self.lib.alq_prof_num_classes()):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            check(self.lib.alq_prof_read(self._ctx, c, C.byref(ms), C.byref(n), C.byref(fl)))
            out[self.lib.alq_prof_class_name(c).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value)
        return out


# ------------------------------------------------------------------------------------------
def _is_extended(layer_dict):
    first = next(iter(layer_dict.values()))
    return isinstance(first[0], str)


def translate_layers(layer_dict, in_shape, skips=()):
    """Reference layer dict (either schema) -> list of dicts in alq_layer_t terms + bookkeeping.

    `in_shape` is the placeholder shape without batch, channels last: (H, W, C) for `NN.CNN`
    / 2-D `NN_extended.CNN`, (D, H, W, C) for 3-D.  Pure host logic (unit-tested on CPU)."""
    ext = _is_extended(layer_dict)
    nd = len(in_shape) - 1
    if nd not in (2, 3):
        raise ValueError('input must be [H,W,C] or [D,H,W,C], got %r' % (in_shape,))
    names = list(layer_dict.keys())

    def pad3(v, fill=1):
        v = list(v)
        return [fill] * (3 - len(v)) + v

    skip_src = {}
    for sk in skips:
        src, dsts, kind = sk
        if kind != 'con':
            raise NotImplementedError("skip type %r: only 'con' is on the scored path" % kind)
        for d in dsts:
            if d in skip_src:
                raise NotImplementedError('layer %d has more than one skip source' % d)
            skip_src[d] = src
    out = []
    for i, name in enumerate(names):
        spec = layer_dict[name]
        last = i == len(names) - 1
        if ext:
            ltype, lspec = spec[0], spec[1]
            order = spec[2] if len(spec) > 2 else 'M'
            if 'B' in order:
                raise NotImplementedError('batch-norm op in layer %s is outside the scored path' % name)
            if order.replace('A', '').replace('M', '') or not order.startswith('M'):
                raise NotImplementedError('op order %r of layer %s' % (order, name))
            relu = 1 if 'A' in order else 0
        else:
            ltype = spec[1]
            relu = 1 if (ltype == 'conv' or (ltype == 'fc' and not last)) else 0
        d = dict(name=name, relu=relu, skip_src=skip_src.get(i, -1), k=[1, 1, 1], s=[1, 1, 1], cout=0)
        if ltype == 'conv':
            d['type'] = ALQ_CONV
            if ext:
                d['cout'] = int(lspec[0])
                d['k'] = pad3(lspec[1])
                if len(lspec) > 2:
                    d['s'] = pad3(lspec[2])
            else:
                d['cout'] = int(spec[0])
                d['k'] = pad3(spec[2])
        elif ltype == 'conv_transpose':
            d['type'] = ALQ_CONVT
            d['cout'] = int(lspec[0])
            d['k'] = pad3(lspec[1])
            d['s'] = pad3(lspec[2])
        elif ltype == 'pool':
            d['type'] = ALQ_POOL
            if ext:
                d['k'] = pad3(lspec)
                d['s'] = pad3(lspec)
            else:
                w, s = spec[0]          # max_pool(x, pool_size[0], pool_size[1]), NN.py:333-335
                d['k'] = pad3([w] * nd)
                d['s'] = pad3([s] * nd)
        elif ltype == 'fc':
            d['type'] = ALQ_FC
            d['cout'] = int(lspec[0]) if ext else int(spec[0])
        else:
            raise ValueError("layer type %r" % (ltype,))
        out.append(d)
    return out


def tf_param_shapes(layers, in_shape):
    """TF variable shapes [(name, W_shape, b_shape)] of the parameterised layers, creation order
    (NN.py:272-277,313-318; NN_extended.py:397-404,436-441,555-561)."""
    nd = len(in_shape) - 1
    spatial = list(in_shape[:-1])
    ch = in_shape[-1]
    chans = {}
    flat = None
    out = []
    for i, d in enumerate(layers):
        if d['skip_src'] >= 0:
            ch += chans[d['skip_src']]
        if d['type'] == ALQ_CONV:
            out.append((d['name'], tuple(d['k'][3 - nd:]) + (ch, d['cout']), (d['cout'],)))
            ch = d['cout']
        elif d['type'] == ALQ_CONVT:
            out.append((d['name'], tuple(d['k'][3 - nd:]) + (d['cout'], ch), (d['cout'],)))
            spatial = [a * b for a, b in zip(spatial, d['s'][3 - nd:])]
            ch = d['cout']
        elif d['type'] == ALQ_POOL:
            spatial = [-(-a // b) for a, b in zip(spatial, d['s'][3 - nd:])]
        else:
            if flat is None:
                flat = int(np.prod(spatial)) * ch
            out.append((d['name'], (d['cout'], flat), (d['cout'], 1)))
            flat = d['cout']
        chans[i] = ch
    return out


class DeviceModel(object):
    """The reference model protocol (`x, keep_prob, posteriors, prediction, feature_layer,
    grad_posts, var_dict, dropout_rate`) over a libalq model."""

    def __init__(self, sess, layer_dict, in_shape, skips=(), feature_layer=None, dropout=None,
                 max_batch=256, name='model'):
        self.sess = sess
        self.lib = sess.lib
        self.name = name
        self.layer_dict = layer_dict
        self.in_shape = tuple(int(v) for v in in_shape)
        self.skips = [list(s) for s in skips]
        self.max_batch = int(max_batch)
        self.layers = translate_layers(layer_dict, self.in_shape, skips)
        self.param_shapes = tf_param_shapes(self.layers, self.in_shape)
        self.var_names = [p[0] for p in self.param_shapes]
        nd = len(self.in_shape) - 1
        dims = [1] * (3 - nd) + list(self.in_shape[:-1]) + [self.in_shape[-1]]
        arr = (LayerT * len(self.layers))()
        for i, d in enumerate(self.layers):
            arr[i].type = d['type']
            arr[i].cout = d['cout']
            arr[i].k[:] = d['k']
            arr[i].s[:] = d['s']
            arr[i].relu = d['relu']
            arr[i].skip_src = d['skip_src']
        self._m = C.c_void_p()
        cd = (C.c_int32 * 4)(*dims)
        self._create_args = (arr, len(self.layers), cd)
        check(self.lib.alq_model_create(sess.ctx, arr, len(self.layers), cd, self.max_batch, C.byref(self._m)))
        _track(self)
        self.max_batch = int(self.lib.alq_model_max_batch(self._m))      # may be below the request: 32-bit tensor offsets (alq.h)
        # extra scoring pipelines (fisher_device): created at the first call that has that many device passes to run.  Default
        # since round 6: two (ALQ_LANES=1: one pipeline; up to 4: three measured no faster than two, four slower): outputs are
        # bit-identical; each costs a set of workspaces
        self.lanes = max(1, min(4, int(os.environ.get('ALQ_LANES', '2'))))
        self._xlanes = []                 # the extra pipelines (lanes - 1 of them once a call spans that many passes)
        self._create_env = {k: v for k, v in os.environ.items() if k.startswith('ALQ_')}     # engine switches are read at creation
        self.L = self.lib.alq_model_num_param_layers(self._m)
        self.nclass = self.layers[-1]['cout']
        self.elems_per_patch = int(np.prod(self.in_shape))
        # reference-style attributes
        self.x = Handle('x')
        self.keep_prob = Handle('keep_prob')
        if dropout:
            self.dropout_layers, self.dropout_rate = dropout[0], dropout[1]
        else:
            self.dropout_layers, self.dropout_rate = [], 1.
        self.posteriors = Handle('posteriors')
        self.prediction = Handle('prediction')
        self.feature_idx = feature_layer
        fdim = 0
        if feature_layer is not None:
            e = C.c_int64()
            check(self.lib.alq_model_layer_out_elems(self._m, int(feature_layer), C.byref(e)))
            fdim = e.value
        self.feature_dim = fdim
        self.feature_layer = Handle('feature_layer', (fdim,))
        for h in (self.posteriors, self.prediction, self.feature_layer):
            h.model = self
        # 2L opaque entries per class, so that len(model.grad_posts['1'])/2 == L (PW_NNAL.py:751)
        self.var_dict = OrderedDict((n, None) for n in self.var_names)
        self.grad_layers = []
        self._build_grad_handles()
        self.y_ = Handle('y_')
        self.train_step = None            # get_optimizer() creates it (NN.py:557-615)
        self._opt = None
        self._weights_version = 0         # bumped by every set_weights: the optimiser's device copy follows it
        self._drop_calls = 0
        self.num_params = int(self.lib.alq_model_num_params(self._m))
        self._feature_perm = self._feature_permutation()

    # -- gradients of the log-posteriors (get_gradients, NN.py:621-645) -------------------
    def _build_grad_handles(self):
        """2L' opaque entries per class - L' = the layers of `grad_layers` (all when empty) - so that
        len(model.grad_posts['1'])/2 is the A-matrix size PW_NNAL.gen_A_matrices reads (PW_NNAL.py:751)."""
        names = list(self.grad_layers) if len(self.grad_layers) else list(self.var_names)
        for nme in names:
            if nme not in self.var_names:
                raise KeyError('grad layer %r is not a parameterised layer of the model' % (nme,))
        self.grad_layer_idx = [self.var_names.index(nme) for nme in names]
        self.grad_posts = {}
        for j in range(self.nclass):
            hs = []
            for t in self.grad_layer_idx:
                for part in ('W', 'b'):
                    h = Handle('grad_%d_%s_%s' % (j, self.var_names[t], part))
                    h.model, h.cls = self, j
                    hs.append(h)
            self.grad_posts[str(j)] = hs

    def get_gradients(self, grad_layers=[]):
        """NN.py:621-645 / NN_extended.py:1011-1035: the gradient lists cover `grad_layers` (all layers when empty)."""
        self.grad_layers = list(grad_layers)
        self._build_grad_handles()

    def unflatten(self, vec, layers=None):
        """Flat parameter-order vector [W_0, b_0, W_1, ...] -> list of arrays in TF variable shapes (of `layers`)."""
        out, off = [], 0
        keep = set(range(self.L)) if layers is None else set(layers)
        for t, (name, wshape, bshape) in enumerate(self.param_shapes):
            nw, nb = int(np.prod(wshape)), int(np.prod(bshape))
            if t in keep:
                out.append(vec[off:off + nw].reshape(wshape))
                out.append(vec[off + nw:off + nw + nb].reshape(bshape))
            off += nw + nb
        return out

    def flat_params(self):
        return np.concatenate([np.concatenate([np.asarray(W, np.float32).ravel(), np.asarray(b, np.float32).ravel()])
                               for W, b in self.var_dict.values()])

    def set_flat_params(self, vec):
        arrs = self.unflatten(np.asarray(vec, dtype=np.float32))
        self.set_weights({n: [arrs[2 * t], arrs[2 * t + 1]] for t, n in enumerate(self.var_names)})

    def _drop_args(self, keep_prob, seed):
        kp = float(keep_prob)
        lay = [int(v) for v in self.dropout_layers] if kp < 1. else []
        arr = (C.c_int32 * max(len(lay), 1))(*lay)
        if seed is None:
            # the reference's tf.nn.dropout draws a fresh mask per sess.run from TF's stream; here: a fresh seed per
            # call from the global NumPy stream (reproducible under np.random.seed like the rest of the query code)
            seed = int(np.random.randint(0, 2 ** 31 - 1)) if kp < 1. else 0
        return kp, arr, len(lay), int(seed)

    def param_grads_device(self, t, n, mode, cls=0, labels=None, loss_scale=1., keep_prob=1., seed=None, first_sample=0,
                           per_sample=True, want_post=False, want_loss=False):
        """alq_param_grads on n device patches (n <= max_batch): mode 0 = gradients of log posteriors[cls, .] per sample,
        mode 1 = gradient of loss_scale * sum CE.  Returns (grads [n, P] or [P], post [c, n] or None, loss or None)."""
        torch = self.sess.torch
        self.sess.bind_stream()
        if n > self.max_batch:
            raise ValueError('%d patches exceed max_batch = %d' % (n, self.max_batch))
        kp, arr, nl, seed = self._drop_args(keep_prob, seed)
        g = self.sess.empty((n, self.num_params) if per_sample else (self.num_params,), torch.float32)
        post = self.sess.empty((self.nclass, n), torch.float32) if want_post else None
        loss = self.sess.empty((1,), torch.float64) if want_loss else None
        lab = None
        if mode == 1:
            lab = labels if isinstance(labels, torch.Tensor) else self.sess.to_device(np.asarray(labels, dtype=np.int32), torch.int32)
        check(self.lib.alq_param_grads(
            self._m, C.c_void_p(t.data_ptr()), n, int(mode), int(cls), C.c_void_p(lab.data_ptr()) if lab is not None else None,
            float(loss_scale), kp, seed, int(first_sample), arr, nl, 1 if per_sample else 0, C.c_void_p(g.data_ptr()),
            C.c_void_p(post.data_ptr()) if post is not None else None, C.c_void_p(loss.data_ptr()) if loss is not None else None))
        return g, post, loss

    def grad_log_post(self, x, j, keep_prob=1.):
        """`sess.run(model.grad_posts[str(j)], {x: batch})`: gradients of log posteriors[j, 0] - sample 0 of the batch,
        like the reference's graph node (NN.py:639-645) - w.r.t. the variables of `grad_layers`, TF shapes."""
        t, n = self._as_device_batch(x)
        g, _, _ = self.param_grads_device(t, 1, 0, cls=j, keep_prob=keep_prob)
        return self.unflatten(g[0].cpu().numpy(), self.grad_layer_idx)

    # -- training step (get_optimizer / train_step, NN.py:557-615) --------------------------
    def get_optimizer(self, learning_rate, train_layers=[], optimizer_name='SGD'):
        """Mean softmax cross-entropy + SGD or Adam on all layers or on `train_layers` (NN.py:583-615)."""
        if optimizer_name not in ('SGD', 'Adam'):
            raise NotImplementedError('optimizer %r (NN.py:591-615 knows SGD and Adam)' % (optimizer_name,))
        for nme in train_layers:
            if nme not in self.var_names:
                raise KeyError('train layer %r' % (nme,))
        self.train_layers = list(train_layers)
        self.train_step = Handle('train_step')
        self.train_step.model = self
        self._opt = dict(name=optimizer_name, lr=float(learning_rate), t=0, theta=None, m=None, v=None)

    def _train_mask(self):
        """1 on the parameters of `train_layers` (all when empty), flat order."""
        if not self.train_layers:
            return None
        mask = np.zeros(self.num_params, dtype=np.float32)
        off = 0
        for name, wshape, bshape in self.param_shapes:
            cnt = int(np.prod(wshape)) + int(np.prod(bshape))
            if name in self.train_layers:
                mask[off:off + cnt] = 1.
            off += cnt
        return mask

    def train_on_batch(self, x, y_onehot, keep_prob=1., seed=None):
        """One `sess.run(model.train_step, {x, y_, keep_prob})`: gradient of the batch-mean cross-entropy (summed over
        device passes of max_batch patches), one optimiser step on the device, weights repacked.  y_onehot: [c, n]
        like the reference's hot_labels (PW_AL.py:1064-1067); an all-zero column is an unlabelled sample.
        Returns the batch-mean loss before the step."""
        if self._opt is None:
            raise RuntimeError('get_optimizer() has not been called (NN.py:1354)')
        torch = self.sess.torch
        t, n = self._as_device_batch(x)
        y = np.asarray(y_onehot)
        if y.shape != (self.nclass, n):
            raise ValueError('labels must be [%d, %d] one-hot columns, got %r' % (self.nclass, n, y.shape))
        lab = np.where(y.sum(0) > 0, y.argmax(0), -1).astype(np.int32)
        kp, _, _, seed = self._drop_args(keep_prob, seed)
        labd = self.sess.to_device(lab, torch.int32)
        gsum = torch.zeros((self.num_params,), dtype=torch.float32, device=self.sess.device)
        loss = 0.
        for a in range(0, n, self.max_batch):
            b = min(n, a + self.max_batch)
            g, _, l = self.param_grads_device(t[a:b], b - a, 1, labels=labd[a:b], loss_scale=1. / n, keep_prob=kp,
                                              seed=seed, first_sample=a, per_sample=False, want_loss=True)
            gsum += g
            loss += float(l.item()) * (b - a) / n
        o = self._opt
        if o['theta'] is None or o.get('version') != self._weights_version:
            # the TF variables are the single state of the reference: weights loaded or assigned since the last step
            # (set_weights / load_weights / perform_assign_ops) are what the next step updates; Adam's slots persist
            fresh = o['theta'] is None
            o['theta'] = self.sess.to_device(self.flat_params(), torch.float32)
        else:
            fresh = False
        if fresh:
            o['m'] = torch.zeros_like(o['theta'])
            o['v'] = torch.zeros_like(o['theta'])
            tm = self._train_mask()
            o['mask'] = self.sess.to_device(tm, torch.float32) if tm is not None else None
        if o.get('mask') is not None:
            gsum *= o['mask']
        o['t'] += 1
        P = self.num_params
        self.sess.bind_stream()
        if o['name'] == 'SGD':
            check(self.lib.alq_sgd_step(self.sess.ctx, C.c_void_p(o['theta'].data_ptr()), C.c_void_p(gsum.data_ptr()), P, o['lr']))
        else:
            check(self.lib.alq_adam_step(self.sess.ctx, C.c_void_p(o['theta'].data_ptr()), C.c_void_p(gsum.data_ptr()),
                                         C.c_void_p(o['m'].data_ptr()), C.c_void_p(o['v'].data_ptr()), P, o['lr'],
                                         0.9, 0.999, 1e-8, o['t']))      # a masked-out parameter keeps m = v = 0: its step is 0 / eps = 0
        self.set_flat_params(o['theta'].cpu().numpy())
        o['version'] = self._weights_version
        return loss

    def diagonal_fisher(self, x, labels=None, batch=None):
        """model_utils.diagonal_Fisher (model_utils.py:294-330): mean over samples of the squared gradient of the
        log-likelihood of the sample's label (labels given) or of the model's own prediction (None), per parameter.
        Returns the list of arrays in variable shapes."""
        torch = self.sess.torch
        t, n = self._as_device_batch(x)
        acc = torch.zeros((self.num_params,), dtype=torch.float64, device=self.sess.device)
        if labels is None:
            post, pred, _ = self.forward_device(t, n, want_pred=True)
            labels = pred.cpu().numpy()
        labels = np.asarray(labels).astype(np.int64)
        step = min(self.max_batch, batch or self.max_batch)
        for j in range(self.nclass):
            idx = np.nonzero(labels == j)[0]
            for a in range(0, len(idx), step):
                sel = self.sess.to_device(idx[a:a + step], torch.int64)
                xs = t.reshape(n, -1).index_select(0, sel)
                g, _, _ = self.param_grads_device(xs, int(sel.numel()), 0, cls=j)
                check(self.lib.alq_sq_accum(self.sess.ctx, C.c_void_p(g.data_ptr()), self.num_params, int(sel.numel()),
                                            C.c_void_p(acc.data_ptr())))
        return self.unflatten((acc / max(n, 1)).cpu().numpy())

    # -- weights ---------------------------------------------------------------------------
    def set_weights(self, pars):
        """`pars`: name -> [W, b] in TF layouts (HWIO / DHWIO, transpose [k..,out,in], fc [out,in],
        fc bias [out,1]); the weight interchange of NN.py:390-394 / :508-517 with numpy arrays in
        place of the HDF5 datasets (h5py is not in the image)."""
        for t, (name, wshape, bshape) in enumerate(self.param_shapes):
            W, b = pars[name]
            W = np.ascontiguousarray(np.asarray(W, dtype=np.float32))
            b = np.ascontiguousarray(np.asarray(b, dtype=np.float32))
            if tuple(W.shape) != tuple(wshape) or b.size != int(np.prod(bshape)):
                raise ValueError('layer %s: expected W%s b%s, got W%s b%s' % (name, wshape, bshape, W.shape, b.shape))
            check(self.lib.alq_model_set_weights(self._m, t, W.ctypes.data_as(C.c_void_p),
                                                 b.ctypes.data_as(C.c_void_p)))
            self.var_dict[name] = [W, b]
        self._weights_version += 1

    def load_weights(self, path, session=None):
        """CNN.load_weights (NN.py:396-419; NN_extended.py:708-760): an HDF5 file of the reference - groups per layer, datasets
        `Weight` / `Bias` in TF layouts - when h5py is importable, or the .npz twin (keys '<layer>/Weight', '<layer>/Bias')."""
        from . import weights_io
        self.set_weights(weights_io.read_weights(path, self.var_names))

    def save_weights(self, path):
        """CNN.save_weights (NN.py:379-394): `.h5` -> the reference's HDF5 layout (needs h5py), anything else -> the .npz twin."""
        from . import weights_io
        weights_io.write_weights(path, self.var_dict)

    def add_assign_ops(self):
        """No graph to extend (NN.py:421-458): kept so loop code calls it unchanged."""

    def perform_assign_ops(self, file_path, sess=None):
        """NN.py:462-519: 'init' draws He-normal weights with the global np.random in variable
        creation order (std = sqrt(2/n), zero biases); otherwise loads a weight file."""
        if file_path != 'init':
            return self.load_weights(file_path)
        pars = OrderedDict()
        for name, wshape, bshape in self.param_shapes:
            n = int(np.prod(wshape[:-1])) if len(wshape) > 2 else wshape[1]
            pars[name] = [np.sqrt(2. / n) * np.random.randn(*wshape), np.zeros(bshape)]
        self.set_weights(pars)

    # -- evaluation ------------------------------------------------------------------------
    def _feature_permutation(self):
        """feature_layer of a conv/pool layer is flattened in the reference's order (full axis
        reversal, NN.py:296-301,337-340); the device returns memory order.  Identity after an fc."""
        if self.feature_idx is None:
            return None
        # re-derive the layer's output geometry
        nd = len(self.in_shape) - 1
        spatial = list(self.in_shape[:-1])
        ch = self.in_shape[-1]
        for i, d in enumerate(self.layers):
            if d['type'] == ALQ_CONV:
                ch = d['cout']
            elif d['type'] == ALQ_CONVT:
                spatial = [a * b for a, b in zip(spatial, d['s'][3 - nd:])]
                ch = d['cout']
            elif d['type'] == ALQ_POOL:
                spatial = [-(-a // b) for a, b in zip(spatial, d['s'][3 - nd:])]
            else:
                return None                      # at or after an fc: already a flat vector
            if i == self.feature_idx:
                mem = np.arange(int(np.prod(spatial)) * ch).reshape(spatial + [ch])
                return mem.transpose(*reversed(range(nd + 1))).reshape(-1)
        return None

    def _as_device_batch(self, x):
        torch = self.sess.torch
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.sess.device, dtype=torch.float32).contiguous()
        else:
            t = self.sess.to_device(np.asarray(x), torch.float32)    # placeholder is tf.float32
        n = t.numel() // self.elems_per_patch
        if n * self.elems_per_patch != t.numel():
            raise ValueError('batch of %d elements is not a multiple of the patch size %d' % (t.numel(), self.elems_per_patch))
        return t, n

    def forward_dropout_device(self, t, n, keep_prob, seed=None, first_sample=0, want_pred=False):
        """Posteriors at keep_prob < 1 (MC strategies): alq_forward_dropout over device passes; masks keyed by sample id."""
        torch = self.sess.torch
        self.sess.bind_stream()
        kp, arr, nl, seed = self._drop_args(keep_prob, seed)
        post = self.sess.empty((self.nclass, n), torch.float32)
        pred = self.sess.empty((n,), torch.int64) if want_pred else None
        for a in range(0, n, self.max_batch):
            b = min(n, a + self.max_batch)
            pb = self.sess.empty((self.nclass, b - a), torch.float32)
            check(self.lib.alq_forward_dropout(self._m, C.c_void_p(t.data_ptr() + a * self.elems_per_patch * 4), b - a, kp, seed,
                                               int(first_sample) + a, arr, nl, C.c_void_p(pb.data_ptr()),
                                               C.c_void_p(pred.data_ptr() + a * 8) if want_pred else None))
            post[:, a:b] = pb
        return post, pred

    def forward_device(self, t, n, want_pred=False, want_feat=False, rows=None):
        """t: device fp32 tensor of n patches - or, with `rows` (int64 device tensor [n]), a resident pool whose
        rows `rows` are the patches (alq_forward_rows: no gathered copy on the caller's side).
        Returns device tensors (post [c,n], pred, feat)."""
        torch = self.sess.torch
        self.sess.bind_stream()
        post = self.sess.empty((self.nclass, n), torch.float32)
        pred = self.sess.empty((n,), torch.int64) if want_pred else None
        feat = self.sess.empty((n, self.feature_dim), torch.float32) if want_feat else None
        if rows is not None:
            assert rows.dtype == torch.int64 and rows.is_contiguous() and int(rows.numel()) == n
        # one pipeline: forward-only passes leave no gaps a second one could fill (configs[4]'s filter: 0.3625 s per 200k patches
        # with two pipelines against 0.3601 s with one, same box)
        for a in range(0, n, self.max_batch):
            b = min(n, a + self.max_batch)
            pb = self.sess.empty((self.nclass, b - a), torch.float32)
            outs = (C.c_void_p(pb.data_ptr()),
                    C.c_void_p(pred.data_ptr() + a * 8) if want_pred else None,
                    C.c_void_p(feat.data_ptr() + a * self.feature_dim * 4) if want_feat else None,
                    self.feature_idx if want_feat else -1)
            if rows is None:
                check(self.lib.alq_forward(self._m, C.c_void_p(t.data_ptr() + a * self.elems_per_patch * 4), b - a, *outs))
            else:
                check(self.lib.alq_forward_rows(self._m, C.c_void_p(t.data_ptr()), C.c_void_p(rows.data_ptr() + a * 8),
                                                b - a, *outs))
            post[:, a:b] = pb
        return post, pred, feat

    def forward(self, x, want=('posteriors',), keep_prob=1.):
        t, n = self._as_device_batch(x)
        if float(keep_prob) < 1. and len(self.dropout_layers):
            if 'feature_layer' in want:
                raise NotImplementedError('feature_layer at keep_prob < 1')
            post, pred = self.forward_dropout_device(t, n, keep_prob, want_pred='prediction' in want)
            res = {'posteriors': post.cpu().numpy()}
            if pred is not None:
                res['prediction'] = pred.cpu().numpy()
            return res
        post, pred, feat = self.forward_device(t, n, 'prediction' in want, 'feature_layer' in want)
        res = {'posteriors': post.cpu().numpy()}
        if pred is not None:
            res['prediction'] = pred.cpu().numpy()
        if feat is not None:
            f = feat.cpu().numpy()
            if self._feature_perm is not None:
                f = f[:, self._feature_perm]
            res['feature_layer'] = np.ascontiguousarray(f.T)       # [F, n] like the reference
        return res

    def fisher_device(self, t, n, p1_in=None, diag_load=1e-5, want=('p1', 'g0', 'g1', 'A', 'trace', 'Asum'), rows=None):
        """Device-resident Fisher scoring of n patches (t: device fp32; with `rows`, rows of the resident pool `t`:
        alq_fisher_rows).  Returns device tensors; 'Asum' is the sum over the n patches (fixed summation order per
        launch).  'H' (Shannon entropy of the posteriors, alq_score_entropy) is produced on request."""
        torch = self.sess.torch
        self.sess.bind_stream()
        if rows is not None:
            assert rows.dtype == torch.int64 and rows.is_contiguous() and int(rows.numel()) == n
        L = self.L
        out = {}
        out['p1'] = self.sess.empty((n,), torch.float32) if 'p1' in want else None
        out['g0'] = self.sess.empty((n, L), torch.float64) if 'g0' in want else None
        out['g1'] = self.sess.empty((n, L), torch.float64) if 'g1' in want else None
        out['A'] = self.sess.empty((n, L, L), torch.float64) if 'A' in want else None
        out['trace'] = self.sess.empty((n,), torch.float64) if 'trace' in want else None
        asum = torch.zeros((L, L), dtype=torch.float64, device=self.sess.device) if 'Asum' in want else None
        part = self.sess.empty((L, L), torch.float64) if asum is not None else None

        def ptr(tn, off_elems, itemsize):
            return C.c_void_p(tn.data_ptr() + off_elems * itemsize) if tn is not None else None

        # Device passes of max_batch patches alternate between two PIPELINES (own libalq context = own stream pair, own
        # workspaces, the same weights): the tail of one pass (the last box-filter dot products, finalisation) and its
        # small kernels run beside the next pass's first launches instead of leaving the chip idle.  Each pass is the same
        # launches on the same data whichever pipeline runs it, so every per-patch output is bit-identical to ALQ_LANES=1;
        # the passes' partial sums of A are added in pass order after the join.
        # Pass sizes: an EVEN number of equal passes, so that two pipelines get the same work (100,000 patches at 2047 per pass are 49
        # passes: one pipeline ran 25 of them and the other idled through the last one: 268.0 against 272.7 k patches/s same-box,
        # three rounds).  Per-patch results do not depend on the cut (tests: any batch cut is bit-identical); the cut itself does not
        # depend on the number of pipelines, so the pass-ordered sum of A is the same bits with one, two or three.  (A multiple of 6
        # - equal shares for three pipelines as well - was tried: 50 ... 56 passes score the same within noise on NET-C, but NET-B
        # streams its 168 MB of fc weights once per pass and 16,384 patches became 12 passes instead of 8: ALQ_PASS_MULT=6.)
        step, starts = self.pass_cut(n)
        nl = min(self.lanes, len(starts))
        extra = self._extra_lanes(nl - 1) if nl > 1 else []
        part = self.sess.empty((max(len(starts), 1), L, L), torch.float64) if asum is not None else None
        cur = torch.cuda.current_stream(self.sess.device)
        # with several pipelines whole passes overlap: the per-context side streams (the statistics kernels of a layer beside its
        # contraction) then only add streams competing for the same gaps (274.4 -> 277.8 k patches/s without them, same box,
        # three rounds); a single pipeline keeps its side stream (+3.4 %, round 4).  Same kernels, same results either way.
        side_on = 0 if (extra and not os.environ.get('ALQ_LANES_SIDE_STREAM')) else 1
        check(self.lib.alq_ctx_use_side_stream(self.sess.ctx, side_on))
        for ln in extra:
            check(self.lib.alq_ctx_use_side_stream(ln['sess'].ctx, side_on))
            ln['stream'].wait_stream(cur)          # inputs, output buffers and the weights are ordered on the caller's stream
        for k, a in enumerate(starts):
            b = min(n, a + step)
            outs = (ptr(p1_in, a, 4), float(diag_load), ptr(out['p1'], a, 4), ptr(out['g0'], a * L, 8),
                    ptr(out['g1'], a * L, 8), ptr(out['A'], a * L * L, 8), ptr(out['trace'], a, 8),
                    C.c_void_p(part.data_ptr() + k * L * L * 8) if part is not None else None)

            def launch(m):
                if rows is None:
                    check(self.lib.alq_fisher(m, C.c_void_p(t.data_ptr() + a * self.elems_per_patch * 4), b - a, *outs))
                else:
                    check(self.lib.alq_fisher_rows(m, C.c_void_p(t.data_ptr()), C.c_void_p(rows.data_ptr() + a * 8), b - a, *outs))
            which = k % nl
            if which:
                ln = extra[which - 1]
                with torch.cuda.stream(ln['stream']):
                    ln['sess'].bind_stream()
                    launch(ln['m'])
            else:
                launch(self._m)
        for ln in extra:
            cur.wait_stream(ln['stream'])
        if not side_on:
            check(self.lib.alq_ctx_use_side_stream(self.sess.ctx, 1))
        if asum is not None:
            for k in range(len(starts)):
                asum += part[k]
        out['Asum'] = asum
        if 'H' in want or 'absdev' in want:
            if out['p1'] is None:
                raise ValueError("'H' / 'absdev' need 'p1' in `want`")
            out['H'] = self.sess.empty((n,), torch.float32) if 'H' in want else None
            out['absdev'] = self.sess.empty((n,), torch.float64) if 'absdev' in want else None
            check(self.lib.alq_score_entropy(self.sess.ctx, C.c_void_p(out['p1'].data_ptr()), n,
                                             ptr(out['absdev'], 0, 8), ptr(out['H'], 0, 4)))
        return out

    def _creation_env(self):
        """The ALQ_* engine switches are read when a model is created and when its weights are packed: the second pipeline's
        model is built under the ones the first was created with."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            now = {k: v for k, v in os.environ.items() if k.startswith('ALQ_')}
            for k in now:
                del os.environ[k]
            os.environ.update(self._create_env)
            try:
                yield
            finally:
                for k in self._create_env:
                    os.environ.pop(k, None)
                os.environ.update(now)
        return cm()

    def pass_cut(self, n):
        """How fisher_device cuts n patches into device passes: (patches per pass, first patch of every pass)."""
        step = self.max_batch
        if n > self.max_batch and not os.environ.get('ALQ_NO_PASS_BALANCE'):
            P = -(-n // self.max_batch)
            mult = int(os.environ.get('ALQ_PASS_MULT', '2'))          # (tuning experiments)
            P = -(-P // mult) * mult if P >= mult else P + (P & 1)
            step = -(-n // P)
        return step, list(range(0, n, step))

    @property
    def _lane2(self):
        """The first extra pipeline (None until a call needed one): what the tests of round 5 look at."""
        return self._xlanes[0] if self._xlanes else None

    def _extra_lanes(self, count):
        """`count` extra scoring pipelines of fisher_device: each a libalq context on its own torch stream and a model of the same
        layers on it; their weights follow `set_weights` (the host copies in var_dict are the single state)."""
        torch = self.sess.torch
        while len(self._xlanes) < count:
            stream = torch.cuda.Stream(self.sess.device)
            with torch.cuda.stream(stream):
                sess2 = DeviceSession(self.sess.device.index)
            m = C.c_void_p()
            arr, nl, cd = self._create_args
            with self._creation_env():
                check(self.lib.alq_model_create(sess2.ctx, arr, nl, cd, self.max_batch, C.byref(m)))
            if int(self.lib.alq_model_max_batch(m)) != self.max_batch:
                self.lib.alq_model_destroy(m)
                raise _lib.AlqError('extra pipeline: the library granted another batch size')
            self._xlanes.append(dict(sess=sess2, stream=stream, m=m, version=None))
        for ln in self._xlanes[:count]:
            if ln['version'] != self._weights_version:
                if any(v is None for v in self.var_dict.values()):
                    raise RuntimeError('set_weights() has not been called')
                with self._creation_env():
                    for ti, name in enumerate(self.var_names):
                        W, b = self.var_dict[name]
                        check(self.lib.alq_model_set_weights(ln['m'], ti, W.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
                ln['version'] = self._weights_version
        return self._xlanes[:count]

    def fisher(self, x, p1=None, diag_load=1e-5):
        t, n = self._as_device_batch(x)
        p1_in = None
        if p1 is not None:
            p1_in = self.sess.to_device(np.asarray(p1, dtype=np.float32), self.sess.torch.float32)
        out = self.fisher_device(t, n, p1_in, diag_load)
        return {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}

    def shrunk_class_gradients(self, x):
        """g [n, c, L'] float64: shrink_gradient(d log posteriors[j, sample] / d theta, 'sum') for every sample and class -
        what the reference obtains with one session.run(model.grad_posts[str(j)]) per sample and class plus a host
        reduction (NNAL.py:381-405, NNAL_tools.py:784-796).  Per device pass: alq_param_grads (mode 0, per-sample rows)
        for class j, alq_shrink_sum on the rows; nothing of size |theta| reaches the host.  L' = the layers of
        `grad_layers` (all when empty).  Also returns the posteriors [c, n] of these passes (device fp32)."""
        torch = self.sess.torch
        t, n = self._as_device_batch(x)
        c, L = self.nclass, self.L
        elems = (C.c_int64 * L)(*[int(np.prod(w)) + int(np.prod(b)) for _, w, b in self.param_shapes])
        g = self.sess.empty((n, c, L), torch.float64)
        post = self.sess.empty((c, n), torch.float32)
        tmp = self.sess.empty((min(n, self.max_batch), L), torch.float64)
        flat = t.reshape(n, -1)
        for a in range(0, n, self.max_batch):
            b = min(n, a + self.max_batch)
            for j in range(c):
                rows, pb, _ = self.param_grads_device(flat[a:b], b - a, 0, cls=j, want_post=(j == 0))
                if j == 0:
                    post[:, a:b] = pb
                check(self.lib.alq_shrink_sum(self.sess.ctx, C.c_void_p(rows.data_ptr()), b - a, self.num_params, elems, L,
                                              C.c_void_p(tmp.data_ptr())))
                g[a:b, j, :] = tmp[:b - a]
        idx = list(self.grad_layer_idx)
        if idx != list(range(L)):
            g = g[:, :, idx].contiguous()
        return g, post

    def fisher_classes(self, x, W, diag):
        """A [n, L', L'] float64 = sum_j W[i, j] g_ij g_ij^T + diag[i] I (alq_fisher_classes on shrunk_class_gradients)."""
        torch = self.sess.torch
        g, _ = self.shrunk_class_gradients(x)
        n, c, L = [int(v) for v in g.shape]
        Wd = self.sess.to_device(np.asarray(W, dtype=np.float64).reshape(n, c), torch.float64)
        dd = self.sess.to_device(np.asarray(diag, dtype=np.float64).reshape(n), torch.float64)
        A = self.sess.empty((n, L, L), torch.float64)
        self.sess.bind_stream()
        check(self.lib.alq_fisher_classes(self.sess.ctx, C.c_void_p(g.data_ptr()), C.c_void_p(Wd.data_ptr()), C.c_void_p(dd.data_ptr()),
                                          n, c, L, C.c_void_p(A.data_ptr())))
        return A.cpu().numpy()

    def debug_tensor(self, layer_idx, what, n):
        """Test hook (alq_model_debug_copy): internal tensor of the last forward/fisher call."""
        torch = self.sess.torch
        cap = {0: None, 1: None}
        e = C.c_int64()
        # size query by over-allocation: the largest activation of a patch is bounded by 64x input
        buf = self.sess.empty((n * max(self.elems_per_patch * 64, 1 << 16),), torch.float32)
        check(self.lib.alq_model_debug_copy(self._m, int(layer_idx), int(what), int(n),
                                            C.c_void_p(buf.data_ptr()), C.byref(e)))
        del cap
        return buf[:e.value].cpu().numpy()

    def close(self):
        for ln in getattr(self, '_xlanes', []):
            self.lib.alq_model_destroy(ln['m'])
            ln['sess'].close()
        self._xlanes = []
        if self._m:
            self.lib.alq_model_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
