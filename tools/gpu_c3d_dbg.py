#!/usr/bin/env python3
"""Debug: per-layer differences between the plane-sweep and the two-slot engine (backward of the head conv)."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device  # noqa: E402
from nnal_amd._lib import check  # noqa: E402
from oracle import netspec  # noqa: E402

sess = device.DeviceSession(0)
ld, sk = netspec.net_c()
in_shape = (32, 32, 32, 1)
pars = netspec.he_init(ld, in_shape, seed=14, skips=sk, bias_std=0.05)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
models = {}
for name, env in (('igemm4', {'ALQ_NO_C3D': '1'}), ('c3d', {})):
    os.environ.pop('ALQ_NO_C3D', None)
    os.environ.update(env)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=N)
    m.set_weights(pars)
    models[name] = m
os.environ.pop('ALQ_NO_C3D', None)
x = sess.empty((N, 32 ** 3), torch.float32)
check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, N, 32 ** 3, C.c_void_p(x.data_ptr())))


def grab(m, layer, what, elems):
    buf = sess.empty((elems,), torch.float32)
    n = C.c_int64(0)
    check(sess.lib.alq_model_debug_copy(m._m, layer, what, N, C.c_void_p(buf.data_ptr()), C.byref(n)))
    torch.cuda.synchronize()
    return buf[:n.value].cpu().numpy().copy()


out = {}
for name, m in models.items():
    r = m.fisher_device(x, N, None, 1e-3, want=('p1', 'g0', 'g1'))
    torch.cuda.synchronize()
    d = {k: v.cpu().numpy().copy() for k, v in r.items() if v is not None}
    nl = len(ld)
    # layer indices of NET-C: 0 enc1, 1 pool1, 2 enc2, 3 pool2, 4 bott, 5 up1, 6 dec1, 7 up2, 8 dec2, 9 fc
    d['dout_up2'] = grab(m, 7, 1, N * 32 ** 3 * 8)
    d['dsum_up2'] = grab(m, 7, 3, N * 32 ** 3)
    d['dsum_enc1'] = grab(m, 0, 3, N * 32 ** 3)
    d['dsum_dec2'] = grab(m, 8, 3, N * 32 ** 3)
    out[name] = d
a, b = out['igemm4'], out['c3d']
print('g0 per layer max rel:', np.abs(a['g0'] - b['g0']).max(0) / np.abs(a['g0']).max(0))
for k in ('dout_up2', 'dsum_up2', 'dsum_enc1', 'dsum_dec2'):
    A, B = a[k], b[k]
    print(k, A.shape, 'max|a|', np.abs(A).max(), 'max|b|', np.abs(B).max(), 'max diff', np.abs(A - B).max(), 'nonzero a/b', (A != 0).mean(), (B != 0).mean())
A = a['dout_up2'].reshape(N, 32, 32, 32, 8)
B = b['dout_up2'].reshape(N, 32, 32, 32, 8)
dd = np.abs(A - B)
print('diff by z', dd.max(axis=(0, 2, 3, 4))[:8], '...', dd.max(axis=(0, 2, 3, 4))[-4:])
print('diff by y', dd.max(axis=(0, 1, 3, 4))[:10])
print('diff by x', dd.max(axis=(0, 1, 2, 4))[:10], dd.max(axis=(0, 1, 2, 4))[14:20])
print('diff by c', dd.max(axis=(0, 1, 2, 3)))
print('diff by patch (nonzero):', [(i, float('%.2e' % v)) for i, v in enumerate(dd.max(axis=(1, 2, 3, 4))) if v > 1e-6][:40])
for k in ('dsum_up2', 'dsum_enc1'):
    d2 = np.abs(a[k] - b[k]).reshape(N, 32, 32, 32)
    print(k, 'by patch:', [(i, float('%.2e' % v)) for i, v in enumerate(d2.max(axis=(1, 2, 3))) if v > 1e-6][:40])
    bad = np.argwhere(d2 > 1e-6)
    if len(bad):
        print('  z set', sorted(set(bad[:, 1]))[:40], 'y set', sorted(set(bad[:, 2]))[:40], 'x set', sorted(set(bad[:, 3]))[:40])
bad = np.argwhere(dd > 1e-6)
if len(bad):
    print('dout bad: patches', sorted(set(bad[:, 0]))[:20], 'z', sorted(set(bad[:, 1]))[:40], 'y', sorted(set(bad[:, 2]))[:40], 'x', sorted(set(bad[:, 3]))[:40], 'c', sorted(set(bad[:, 4])))
