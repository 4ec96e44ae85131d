#!/usr/bin/env python3
"""Fused enc2 backward (e3d.hip) against the three launches of round 4: where do the channel-sum fields differ (GPU box)."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sess = device.DeviceSession(0)
ld, sk = netspec.net_c()
in_shape = (32, 32, 32, 1)
pars = netspec.he_init(ld, in_shape, seed=14, skips=sk, bias_std=0.05)
x = sess.empty((n, 32 ** 3), torch.float32)
check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
outs = []
for env in ({}, {'ALQ_NO_E3D': '1'}, {'ALQ_NO_E3D': '1', 'ALQ_NO_F16X2': '1'}):
    for k in ('ALQ_NO_E3D', 'ALQ_NO_F16X2'):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=n)
    m.set_weights(pars)
    m.fisher_device(x, n, None, 1e-3, want=('p1',))
    outs.append({'e2': m.debug_tensor(2, 3, n).reshape(n, 16, 16, 16), 'e1': m.debug_tensor(0, 3, n).reshape(n, 32, 32, 32)})
    print(env, 'fused', sess.lib.alq_model_engine_info(m._m, 9))
    m.close()
a, b, c = outs
for k in ('e2', 'e1'):
    for nm, o in (('fused vs 3 launches', b), ('fused vs bf16x3', c)):
        e = np.abs(a[k] - o[k])
        print(k, nm, 'max err %.3e  max ref %.3e  rms err %.3e rms ref %.3e' % (e.max(), np.abs(o[k]).max(), np.sqrt((e ** 2).mean()), np.sqrt((o[k] ** 2).mean())))
    e = np.abs(b[k] - c[k])
    print(k, '3 launches vs bf16x3: max err %.3e' % e.max())
e = np.abs(a['e1'] - c['e1'])
idx = np.unravel_index(np.argsort(-e.reshape(-1))[:8], e.shape)
print('largest errors at (patch, z, y, x):', list(zip(*[i.tolist() for i in idx])))
for ax, nm in ((1, 'z'), (2, 'y'), (3, 'x')):
    prof = e.max(axis=tuple(i for i in range(4) if i != ax))
    print('max err by', nm, np.array2string(prof, precision=1))
