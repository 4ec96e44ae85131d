"""A small lazy-graph `tensorflow` stand-in, used ONLY by the golden generators in the build container.

Why: TensorFlow is absent, so round 1's goldens let the oracle's own restatement of the graph answer
every `sess.run`.  With this module registered as `tensorflow`, the reference's OWN graph-construction code -
`NN.CNN.__init__` / `add_conv` / `add_fc` / `add_pool` / `get_gradients` / `get_optimizer` (NN.py:56-645) and
`NN_extended.CNN.__init__` / `add_layer` / `add_conv` / `add_conv_transpose` / `combine_layer_outputs` /
`get_gradients` (NN_extended.py:65-601, 1011-1216) - runs unmodified: layer order, variable order, flatten order,
op-order strings, skip concatenation order and the gradient node definitions are EXECUTED, not restated.
What stays a restatement (and therefore "parity unpinned"): the op kernels themselves.  Every `tf.nn.*` call lands in
`oracle.tfops` (torch-CPU) - the same functions the oracle's own graph uses, so the two can be compared bit for bit.

Mechanics: every op builds a Node (function + inputs); a Node's static shape comes from evaluating it on two probe
feeds that differ only in the size given to unknown (None) dimensions; `Session.run` evaluates Nodes recursively with a
per-run memo; Variables are torch tensors with requires_grad, so `tf.gradients` is `torch.autograd.grad` on the
evaluated graph.  Only what the two constructors and the query path touch is implemented; anything else raises."""
import contextlib
import types

import numpy as np
import torch

from oracle import tfops

float32 = 'float32'
float64 = 'float64'
int32 = 'int32'
int64 = 'int64'
bool = 'bool'          # noqa: A001  (tf.bool)
AUTO_REUSE = object()
_DT = {'float32': torch.float32, 'float64': torch.float64, 'int32': torch.int32, 'int64': torch.int64, 'bool': torch.bool}
_PROBE = (2, 3)


class Dimension(object):
    def __init__(self, v):
        self.value = None if v is None else int(v)

    def _v(self, o):
        return o.value if isinstance(o, Dimension) else o

    def __mul__(self, o):
        o = self._v(o)
        return Dimension(None if (self.value is None or o is None) else self.value * o)
    __rmul__ = __mul__

    def __eq__(self, o):
        return self.value == self._v(o)

    def __ne__(self, o):
        return not self.__eq__(o)

    def __hash__(self):
        return hash(self.value)

    def __int__(self):
        return int(self.value)
    __index__ = __int__

    def __repr__(self):
        return 'Dimension(%r)' % (self.value,)


class TensorShape(object):
    def __init__(self, dims):
        self.dims = [d if isinstance(d, Dimension) else Dimension(d) for d in dims]

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return TensorShape(self.dims[i])
        return self.dims[i]

    def as_list(self):
        return [d.value for d in self.dims]

    def __repr__(self):
        return 'TensorShape(%r)' % (self.as_list(),)


class _Graph(object):
    def __init__(self):
        self.trainable = []
        self.all_vars = []
        self.scope = []
        self.store = {}


_G = _Graph()


def reset_default_graph():
    global _G
    _G = _Graph()


def _val(x, env):
    return x._eval(env) if isinstance(x, Node) else x


class Node(object):
    def __init__(self, fn, inputs=(), name=None):
        self.fn, self.inputs, self.name = fn, list(inputs), name
        self._probe = [self._run_probe(k) for k in range(2)]
        pa, pb = self._probe
        if isinstance(pa, torch.Tensor):
            self.shape = TensorShape([a if a == b else None for a, b in zip(pa.shape, pb.shape)])
            self.dtype = pa.dtype
        else:
            self.shape = TensorShape([])

    def _run_probe(self, k):
        return self.fn(*[(i._probe[k] if isinstance(i, Node) else i) for i in self.inputs])

    def get_shape(self):
        return self.shape

    def _eval(self, env):
        if id(self) not in env:
            env[id(self)] = self.fn(*[_val(i, env) for i in self.inputs])
        return env[id(self)]

    # operators used by the reference's graph code
    def __add__(self, o):
        return Node(lambda a, b: a + b, [self, o])

    def __radd__(self, o):
        return Node(lambda a, b: b + a, [self, o])

    def __sub__(self, o):
        return Node(lambda a, b: a - b, [self, o])

    def __rsub__(self, o):
        return Node(lambda a, b: b - a, [self, o])

    def __mul__(self, o):
        return Node(lambda a, b: a * b, [self, o])

    def __rmul__(self, o):
        return Node(lambda a, b: b * a, [self, o])

    def __truediv__(self, o):
        return Node(lambda a, b: a / b, [self, o])

    def __neg__(self):
        return Node(lambda a: -a, [self])

    def __getitem__(self, idx):
        return Node(lambda a: a[idx], [self])

    def __hash__(self):
        return id(self)

    def __eq__(self, o):            # combine_layer_outputs compares nodes by identity (`model.output==sources_output[-1]`)
        return self is o


class Placeholder(Node):
    def __init__(self, dtype, shape=None, name=None):
        self.pdtype, self.pshape = _DT.get(dtype, torch.float32), shape
        super(Placeholder, self).__init__(None, [], name)

    def _run_probe(self, k):
        if self.pshape is None:
            return torch.tensor(1.0 if self.pdtype.is_floating_point else 0).to(self.pdtype)
        return torch.zeros([(_PROBE[k] if d is None else int(d)) for d in self.pshape], dtype=self.pdtype)

    def _eval(self, env):
        if id(self) not in env:
            raise KeyError('placeholder %r was not fed' % (self.name,))
        return env[id(self)]


class Variable(Node):
    def __init__(self, initial_value, trainable=True, name=None, dtype=None):
        init = initial_value._probe[0] if isinstance(initial_value, Node) else torch.as_tensor(np.asarray(initial_value))
        if init.dtype == torch.float64:
            init = init.float()
        self.tensor = init.clone().detach()
        self.trainable = trainable and self.tensor.dtype.is_floating_point
        if self.trainable:
            self.tensor.requires_grad_(True)
        scope = '/'.join(_G.scope)
        self.vname = (scope + '/' if scope else '') + (name or 'Variable')
        super(Variable, self).__init__(None, [], self.vname)
        _G.all_vars.append(self)
        if self.trainable:
            _G.trainable.append(self)

    def _run_probe(self, k):
        return self.tensor

    def _eval(self, env):
        return self.tensor

    def load(self, value):
        with torch.no_grad():
            self.tensor.copy_(torch.as_tensor(np.asarray(value)).to(self.tensor.dtype).reshape(self.tensor.shape))

    def assign(self, value):
        def do(v):
            self.load(v.detach().numpy() if isinstance(v, torch.Tensor) else v)
            return self.tensor
        return Node(do, [value])

    @property
    def initializer(self):
        return Node(lambda: None, [])


def placeholder(dtype, shape=None, name=None):
    return Placeholder(dtype, shape, name)


def get_variable(name, shape=None, initializer=None, regularizer=None, custom_getter=None, trainable=True, dtype=None):
    key = '/'.join(_G.scope + [name])
    if key in _G.store:
        return _G.store[key]
    v = Variable(initializer, trainable=trainable, name=name)
    _G.store[key] = v
    return v


@contextlib.contextmanager
def _scope(name, **kw):
    _G.scope.append(str(name))
    try:
        yield
    finally:
        _G.scope.pop()


variable_scope = _scope
name_scope = _scope


def get_variable_scope():
    return types.SimpleNamespace(name='/'.join(_G.scope), reuse_variables=lambda: None)


def trainable_variables():
    return list(_G.trainable)


def global_variables():
    return list(_G.all_vars)


def global_variables_initializer():
    return Node(lambda: None, [])


def variables_initializer(var_list=None, name=None):
    return Node(lambda: None, [])


def constant(value, dtype=None, shape=None, name=None):
    t = torch.as_tensor(np.asarray(value, dtype=np.float32 if isinstance(value, float) else None))
    if shape is not None:
        t = torch.full([int(s) for s in shape], float(value), dtype=torch.float32)
    return Node(lambda: t, [])


def random_normal(shape, mean=0., stddev=1., dtype=None, seed=None, name=None):
    shp = [int(s) for s in shape]
    t = torch.as_tensor((mean + stddev * np.random.randn(*shp)).astype(np.float32))
    return Node(lambda: t, [])


def zeros(shape, dtype=None, name=None):
    return constant(0., shape=shape)


def ones(shape, dtype=None, name=None):
    return constant(1., shape=shape)


def to_float(x, name=None):
    return Node(lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(a)).float(), [x])


def cast(x, dtype, name=None):
    return Node(lambda a: a.to(_DT[dtype]), [x])


def identity(x, name=None):
    return Node(lambda a: a, [x], name)


def exp(x, name=None):
    return Node(torch.exp, [x])


def log(x, name=None):
    return Node(torch.log, [x])


def add(a, b, name=None):
    return Node(lambda x, y: x + y, [a, b])


def multiply(a, b, name=None):
    return Node(lambda x, y: x * y, [a, b])


def matmul(a, b, name=None):
    return Node(lambda x, y: x @ y, [a, b])


def transpose(x, perm=None, name=None):
    if perm is None:
        return Node(lambda a: a.permute(*reversed(range(a.dim()))), [x], name)
    return Node(lambda a: a.permute(*perm), [x], name)


def reshape(x, shape, name=None):
    return Node(lambda a: a.reshape([int(s.value if isinstance(s, Dimension) else s) for s in shape]), [x], name)


def concat(values, axis, name=None):
    return Node(lambda *v: torch.cat(v, dim=axis), list(values))


def stack(values, axis=0, name=None):
    return Node(lambda *v: torch.stack(v, dim=axis), list(values))


def shape(x, name=None):          # noqa: A001
    return Node(lambda a: torch.as_tensor(list(a.shape)), [x])


def argmax(x, axis=None, name=None, dimension=None):
    ax = axis if axis is not None else (dimension if dimension is not None else 0)
    return Node(lambda a: a.argmax(dim=ax), [x], name)


def equal(a, b, name=None):
    return Node(lambda x, y: x == y, [a, b])


def reduce_mean(x, axis=None, name=None):
    return Node(lambda a: a.float().mean() if axis is None else a.float().mean(dim=axis), [x], name)


def reduce_sum(x, axis=None, name=None):
    return Node(lambda a: a.sum() if axis is None else a.sum(dim=axis), [x], name)


def gradients(ys, xs, name=None):
    """tf.gradients(y, xs): list of Nodes, one per x (a None gradient becomes zeros, like the oracle's restatement)."""
    xs = list(xs)

    def grads(y, *vals):
        g = torch.autograd.grad(y, list(vals), allow_unused=True, retain_graph=True)
        return [torch.zeros_like(v) if gi is None else gi for v, gi in zip(vals, g)]
    allg = Node(grads, [ys] + xs)
    return [Node(lambda gl, k=k: gl[k], [allg]) for k in range(len(xs))]


class _NN(object):
    @staticmethod
    def conv2d(x, W, strides, padding, name=None):
        assert padding == 'SAME' and strides[0] == 1 and strides[-1] == 1
        return Node(lambda a, w: tfops.conv_same(a, w, torch.zeros(1, dtype=a.dtype), list(strides[1:-1])), [x, W])
    conv3d = conv2d

    @staticmethod
    def conv2d_transpose(x, W, output_shape, strides, padding='SAME', name=None):
        assert padding == 'SAME'
        return Node(lambda a, w: tfops.conv_transpose_same(a, w, torch.zeros(1, dtype=a.dtype), list(strides[1:-1])), [x, W])
    conv3d_transpose = conv2d_transpose

    @staticmethod
    def max_pool(x, ksize, strides, padding, name=None):
        assert padding == 'SAME' and list(ksize) == list(strides)
        return Node(lambda a: tfops.max_pool_same(a, list(ksize[1:-1]), list(strides[1:-1])), [x])
    max_pool3d = max_pool

    @staticmethod
    def relu(x, name=None):
        return Node(torch.relu, [x])

    @staticmethod
    def softmax(x, name=None, axis=-1):
        return Node(lambda a: torch.softmax(a, dim=-1), [x], name)

    @staticmethod
    def dropout(x, keep_prob, name=None):
        def f(a, kp):
            if float(kp) >= 1.:
                return a
            raise NotImplementedError('tf.nn.dropout at keep_prob < 1: TensorFlow\'s random stream is not reproducible here')
        return Node(f, [x, keep_prob])

    @staticmethod
    def softmax_cross_entropy_with_logits(labels=None, logits=None, name=None):
        return Node(lambda y, z: -(y * torch.log_softmax(z, dim=-1)).sum(dim=-1), [labels, logits])


nn = _NN()


class _Opt(object):
    """GradientDescentOptimizer / AdamOptimizer .minimize(loss, var_list): an op Node that applies one step (TF-1.x
    update rules as documented; the training step is outside what the goldens of this module pin)."""

    def __init__(self, kind, lr, **kw):
        self.kind, self.lr, self.t, self.m, self.v = kind, lr, 0, {}, {}

    def minimize(self, loss, var_list=None, global_step=None, name=None):
        vs = list(var_list) if var_list is not None else trainable_variables()
        gs = gradients(loss, vs)

        def step(*g):
            self.t += 1
            lr = float(self.lr._probe[0]) if isinstance(self.lr, Node) else float(self.lr)
            with torch.no_grad():
                for v, gi in zip(vs, g):
                    if self.kind == 'sgd':
                        v.tensor -= lr * gi
                    else:
                        m = self.m.setdefault(id(v), torch.zeros_like(gi))
                        s = self.v.setdefault(id(v), torch.zeros_like(gi))
                        m.mul_(0.9).add_(0.1 * gi)
                        s.mul_(0.999).add_(0.001 * gi * gi)
                        lr_t = lr * np.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
                        v.tensor -= lr_t * m / (s.sqrt() + 1e-8)
            return None
        op = Node.__new__(Node)
        op.fn, op.inputs, op.name, op.shape, op._probe = step, list(gs), name, TensorShape([]), [None, None]
        return op


train = types.SimpleNamespace(
    GradientDescentOptimizer=lambda lr, **kw: _Opt('sgd', lr),
    AdamOptimizer=lambda lr, **kw: _Opt('adam', lr),
    RMSPropOptimizer=lambda lr, **kw: _Opt('sgd', lr),
)
summary = types.SimpleNamespace(scalar=lambda *a, **k: None, FileWriter=lambda *a, **k: None, merge_all=lambda: None)
GraphKeys = types.SimpleNamespace(UPDATE_OPS='update_ops', TRAINABLE_VARIABLES='trainable', GLOBAL_VARIABLES='global',
                                  REGULARIZATION_LOSSES='reg')


def get_collection(key, scope=None):
    return []


class Session(object):
    def __init__(self, *a, **k):
        self.graph = types.SimpleNamespace(finalize=lambda: None)

    def run(self, fetches, feed_dict=None):
        env = {}
        for k, v in (feed_dict or {}).items():
            t = torch.as_tensor(np.asarray(v))
            if t.dtype == torch.float64:
                t = t.float()                      # the placeholders are tf.float32
            env[id(k)] = t.to(k.pdtype) if isinstance(k, Placeholder) else t

        def out(f):
            if isinstance(f, (list, tuple)):
                return [out(g) for g in f]
            v = f._eval(env)
            return v.detach().numpy() if isinstance(v, torch.Tensor) else v
        with torch.enable_grad():
            return out(fetches)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
