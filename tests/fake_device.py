"""CPU stand-ins for device.DeviceSession / device.DeviceModel, backed by the oracle: they give the host logic above
the C ABI (al_loop's sharded rounds, pool_shard) something to run on in the world_size-2 gloo tests, where no GPU and
no libalq compute exist.  TEST INFRASTRUCTURE: per-patch results are pure functions of the patch (as on the device,
whose engines are batch-invariant), which is the property the sharded-equals-single test rests on."""
import numpy as np
import torch

from oracle import alpath
from oracle.model import OracleModel, OracleSession


class FakeSession(object):
    torch = torch
    comm_world = 0

    def to_device(self, arr, dtype):
        return torch.as_tensor(np.ascontiguousarray(arr)).to(dtype)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def uncertainty_filter(self, posts, B, with_keys=False):
        key = np.abs(posts.numpy().astype(np.float64) - 0.5)
        idx = np.argsort(key, kind='stable')[:B].astype(np.int64)
        if with_keys:
            return torch.as_tensor(idx), torch.as_tensor(key[idx])
        return torch.as_tensor(idx)

    def bind_stream(self):
        pass

    def topk_smallest(self, keys, B):
        k = keys.numpy().view(np.uint64)                      # bit-pattern order like alq_topk_uncertain
        return torch.as_tensor(np.lexsort((np.arange(len(k)), k))[:int(B)].astype(np.int64))

    def run(self, fetch, feed_dict=None):
        assert fetch.name == 'train_step'
        m = self.model
        return m.train_on_batch(torch.as_tensor(np.asarray(feed_dict[m.x], dtype=np.float32)), feed_dict[m.y_])


class FakeVolumes(object):
    """patch_utils.DeviceVolumes on the CPU (oracle gather + the two normalisation rules)."""

    def __init__(self, sess, padded_imgs):
        self.vols = [np.asarray(v) for v in padded_imgs]
        self.m = len(self.vols)

    def gather(self, inds, patch_shape, stats=None, quirk=2, out_f64=False):
        p = alpath.get_patches(self.vols, np.asarray(inds, dtype=np.int64), tuple(patch_shape), True) if len(inds) else \
            np.zeros((0, patch_shape[0], patch_shape[1], self.m * patch_shape[2]))
        if quirk != 2 and len(inds):
            st = np.asarray(stats, dtype=np.float64).reshape(-1)[:2 * self.m].reshape(self.m, 2)
            d3 = patch_shape[2]
            for j in range(self.m):
                sl = slice(j, j + 1) if quirk == 1 else slice(j * d3, (j + 1) * d3)
                p[:, :, :, sl] = (p[:, :, :, sl] - st[j, 0]) / st[j, 1]
        return torch.as_tensor(p if out_f64 else p.astype(np.float32))


class _H(object):
    def __init__(self, name):
        self.name = name


class FakeModel(object):
    dropout_rate = 1.
    dropout_layers = ()
    grad_layers = ()
    x, keep_prob, y_, train_step = _H('x'), _H('keep_prob'), _H('y_'), _H('train_step')

    def add_assign_ops(self):
        pass

    def perform_assign_ops(self, path, sess=None):
        if path != 'init':
            self.load_weights(path)

    def __init__(self, ld, in_shape, pars, skips=(), lr=None):
        from oracle.train import OracleOptimizer
        self.om = OracleModel(ld, in_shape, pars, skips=skips)
        self.osess = OracleSession(self.om)
        self.in_shape = tuple(in_shape)
        self.L = self.om.nlayers_par
        self.opt = OracleOptimizer(self.om, lr, (), 'SGD') if lr else None

    def train_on_batch(self, x, y_onehot, keep_prob=1., seed=None):
        return self.opt.step(x.numpy().reshape((-1,) + self.in_shape), y_onehot)

    def weights(self):
        return {n: [w.detach().numpy().copy(), b.detach().numpy().copy()] for n, (w, b) in self.om.params.items()}

    def save_weights(self, path):
        np.savez(path, **{n + '/' + k: v for n, wb in self.weights().items() for k, v in zip(('Weight', 'Bias'), wb)})

    def load_weights(self, path):
        f = np.load(path)
        with torch.no_grad():
            for n, (w, b) in self.om.params.items():
                w.copy_(torch.as_tensor(f[n + '/Weight']))
                b.copy_(torch.as_tensor(f[n + '/Bias']))

    def forward_device(self, t, n, want_pred=False, want_feat=False, rows=None):
        x = t.numpy() if rows is None else t.numpy()[rows.numpy()]
        post = self.om.forward(x.reshape((-1,) + self.in_shape)[:n])['posteriors']
        return torch.as_tensor(post), None, None

    def fisher_device(self, t, n, p1_in=None, diag_load=1e-5, want=('A',), rows=None):
        x = t.numpy() if rows is None else t.numpy()[rows.numpy()]
        x = x.reshape((-1,) + self.in_shape)[:n]

        class E(object):
            pars = {'patch_shape': self.in_shape[:3]}
            nclass = 2
        p = p1_in.numpy().astype(np.float64) if p1_in is not None else self.om.forward(x)['posteriors'][1].astype(np.float64)
        A = np.stack(alpath.gen_A_matrices(E(), self.om, self.osess, x, p, diag_load)) if n else np.zeros((0, self.L, self.L))
        out = {'A': torch.as_tensor(A)}
        if 'p1' in want:
            out['p1'] = torch.as_tensor(p.astype(np.float32))
        if 'Asum' in want:
            out['Asum'] = torch.as_tensor(A.sum(0))
        return out
