#!/usr/bin/env python3
"""enc2 forward + pool2 as one launch (f3d.hip) against the two-slot engine's fp16-pair launch + the pool kernel (ALQ_NO_F3D=1, ALQ_F16_DERIVED_MASK=68 so that
both arms contract fp16 pairs): enc2's output, the pooled tensor, its arg-max, the channel sums through the scores (GPU box)."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402
from nnal_amd._lib import check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sess = device.DeviceSession(0)
ld, sk = netspec.net_c()
in_shape = (32, 32, 32, 1)
pars = netspec.he_init(ld, in_shape, seed=14, skips=sk, bias_std=0.05)
x = sess.empty((n, 32 ** 3), torch.float32)
check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
outs = []
for env in ({}, {'ALQ_NO_F3D': '1', 'ALQ_F16_DERIVED_MASK': '68'}):
    for k in ('ALQ_NO_F3D', 'ALQ_F16_DERIVED_MASK'):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=n)
    m.set_weights(pars)
    r = m.fisher_device(x, n, None, 1e-3, want=('p1', 'g0', 'g1'))
    d = {k: r[k].cpu().numpy() for k in ('p1', 'g0', 'g1')}
    d['enc2'] = m.debug_tensor(2, 0, n).reshape(n, 16, 16, 16, 16)
    d['pool2'] = m.debug_tensor(3, 0, n).reshape(n, 8, 8, 8, 16)
    outs.append(d)
    print(env, 'f3f', sess.lib.alq_model_engine_info(m._m, 12), flush=True)
    m.close()
a, b = outs
for k in ('enc2', 'pool2', 'p1', 'g0', 'g1'):
    e = np.abs(a[k] - b[k])
    print(k, 'max diff', e.max(), 'max ref', np.abs(b[k]).max(), 'finite', np.isfinite(a[k]).all())
# the pooled tensor must be the exact window maximum of the kernel's OWN enc2 output
pm = a['enc2'].reshape(n, 8, 2, 8, 2, 8, 2, 16).max(axis=(2, 4, 6))
print('pool == max of own output:', np.array_equal(pm, a['pool2']))
for key in ('enc2', 'pool2'):
    u = np.abs(a[key] - b[key])
    tol = 1e-5 * np.abs(b[key]).max()
    pb = np.nonzero(u.reshape(n, -1).max(axis=1) > tol)[0]
    print('patches with %s diff:' % key, len(pb), pb[:20].tolist())
    for p in pb[:2]:
        zz = np.nonzero(u[p].max(axis=(1, 2, 3)) > tol)[0]
        yy = np.nonzero(u[p].max(axis=(0, 2, 3)) > tol)[0]
        xx = np.nonzero(u[p].max(axis=(0, 1, 3)) > tol)[0]
        cc = np.nonzero(u[p].max(axis=(0, 1, 2)) > tol)[0]
        print(' patch', p, 'z', zz.tolist(), 'y', yy.tolist(), 'x', xx.tolist(), 'c', cc.tolist())
