#!/usr/bin/env python3
"""up1's backward-data launch on the kernel of t3d8b.hip against the two-slot engine (ALQ_NO_T3D8B=1): bott's cotangent, its channel sums, the scores (GPU box)."""
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
for env in ({}, {'ALQ_NO_T3D8B': '1'}):
    os.environ.pop('ALQ_NO_T3D8B', None)
    os.environ.update(env)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=n)
    m.set_weights(pars)
    r = m.fisher_device(x, n, None, 1e-3, want=('p1', 'g0', 'g1'))
    d = {k: r[k].cpu().numpy() for k in ('p1', 'g0', 'g1')}
    d['bott_dout'] = m.debug_tensor(4, 1, n).reshape(n, 8, 8, 8, 32)
    d['bott_dsum'] = m.debug_tensor(4, 3, n).reshape(n, 8, 8, 8)
    outs.append(d)
    print(env, 't3b launches', sess.lib.alq_model_engine_info(m._m, 8), flush=True)
    m.close()
a, b = outs
for k in ('bott_dout', 'bott_dsum', 'p1', 'g0', 'g1'):
    e = np.abs(a[k] - b[k])
    print(k, 'max diff', e.max(), 'max ref', np.abs(b[k]).max(), 'finite', np.isfinite(a[k]).all())
u = np.abs(a['bott_dout'] - b['bott_dout'])
tol = 1e-5 * np.abs(b['bott_dout']).max()
pb = np.nonzero(u.reshape(n, -1).max(axis=1) > tol)[0]
print('patches with bott_dout diff:', len(pb), pb[:20].tolist())
for p in pb[:3]:
    zz = np.nonzero(u[p].max(axis=(1, 2, 3)) > tol)[0]
    yy = np.nonzero(u[p].max(axis=(0, 2, 3)) > tol)[0]
    xx = np.nonzero(u[p].max(axis=(0, 1, 3)) > tol)[0]
    cc = np.nonzero(u[p].max(axis=(0, 1, 2)) > tol)[0]
    print(' patch', p, 'z', zz.tolist(), 'y', yy.tolist(), 'x', xx.tolist(), 'c', cc.tolist())
if len(pb):
    p = int(pb[0])
    z = int(np.nonzero(u[p].max(axis=(1, 2, 3)) > tol)[0][0])
    y = int(np.nonzero(u[p, z].max(axis=(1, 2)) > tol)[0][0])
    np.set_printoptions(precision=6, linewidth=250, suppress=False)
    print('patch', p, 'z', z, 'y', y)
    print('new x 0..1:\n', a['bott_dout'][p, z, y, :2])
    print('ref:\n', b['bott_dout'][p, z, y, :2])
