#!/usr/bin/env python3
"""Row-sweep conv_transpose engine against the two-slot engine, forward-only pass, layer by layer (GPU box)."""
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
for env in ({}, {'ALQ_NO_T3D': '1'}):
    os.environ.pop('ALQ_NO_T3D', None)
    os.environ.update(env)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=n)
    m.set_weights(pars)
    post = m.forward_device(x, n)[0].cpu().numpy()
    d = {'post': post}
    nn_ = n
    d['up2'] = m.debug_tensor(7, 0, nn_)
    d['dec1'] = m.debug_tensor(6, 0, nn_)
    outs.append(d)
    print(env, 't3f launches', sess.lib.alq_model_engine_info(m._m, 7))
    m.close()
a, b = outs
for k in ('dec1', 'up2', 'post'):
    e = np.abs(a[k] - b[k])
    print(k, 'max diff', e.max(), 'max ref', np.abs(b[k]).max())
e = np.abs(a['post'] - b['post']).max(axis=0)
bad = np.nonzero(e > 2e-6)[0]
print('patches with post diff:', len(bad), bad[:40].tolist())
u = np.abs(a["up2"] - b["up2"]).reshape(n, 32, 32, 32, 8)
pb = np.nonzero(u.reshape(u.shape[0], -1).max(axis=1) > 1e-5)[0]
print('patches with up2 diff:', pb.tolist())
for p in pb[:3]:
    zz = np.nonzero(u[p].max(axis=(1, 2, 3)) > 1e-5)[0]
    yy = np.nonzero(u[p].max(axis=(0, 2, 3)) > 1e-5)[0]
    print(' patch', p, 'z', zz.tolist(), 'y', yy.tolist())
if len(pb):
    p = int(pb[0])
    zz = np.nonzero(u[p].max(axis=(1, 2, 3)) > 1e-5)[0]
    z = int(zz[0])
    yy = np.nonzero(u[p, z].max(axis=(1, 2)) > 1e-5)[0]
    y = int(yy[0])
    A = a['up2'].reshape(n, 32, 32, 32, 8)[p, z, y]
    B = b['up2'].reshape(n, 32, 32, 32, 8)[p, z, y]
    np.set_printoptions(precision=3, linewidth=200, suppress=True)
    print('patch', p, 'z', z, 'y', y, 'bad x:', np.nonzero(np.abs(A - B).max(axis=1) > 1e-5)[0].tolist())
    print('new (co 0..7) at x 0..7:\n', A[:8])
    print('ref:\n', B[:8])
    print('all bad (z, y) of the patch:', [(int(zq), np.nonzero(u[p, zq].max(axis=(1, 2)) > 1e-5)[0].tolist()) for zq in zz])
if len(pb):
    print('new x 24..27:\n', A[24:28])
    print('ref x 24..27:\n', B[24:28])
    Ball = b['up2'].reshape(n, 32, 32, 32, 8)
    seg = A[24:32].reshape(-1)
    d = np.abs(Ball[:, :, :, 24:32, :].reshape(n, 32, 32, -1) - seg[None, None, None, :]).max(axis=-1)
    idx = np.unravel_index(np.argmin(d), d.shape)
    print('closest reference segment (patch, z, y):', idx, 'distance', d[idx])
    Aall = a['up2'].reshape(n, 32, 32, 32, 8)
    print('same row, x 16..23 equal to ref:', np.abs(A[16:24] - B[16:24]).max())
