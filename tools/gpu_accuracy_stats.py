#!/usr/bin/env python3
"""Accuracy of the shipped engines at bench scale as NUMBERS (GPU box): one batch of the bench's pool, default engines and the
A/B arms against the exact-fp32 MFMA engine (alq_debug_set(4, 1)) - patches over 2e-6, patches over north_star's 1e-4, maxima -
and every patch over 1e-4 through the fp64 arbiter (tests/factored_ref.relu_flip_explains), timed.

    python tools/gpu_accuracy_stats.py [n = 2000] [arbitrate = 1]"""
import ctypes as C
import json
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device  # noqa: E402
from nnal_amd._lib import check  # noqa: E402
from oracle import netspec  # noqa: E402
from oracle.model import OracleModel  # noqa: E402
import factored_ref  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
arb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sess = device.DeviceSession(0)
ld, sk = netspec.net_c()
in_shape = (32, 32, 32, 1)
pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
x = sess.empty((n, 32 ** 3), torch.float32)
check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
ARMS = (('default', {}, None), ('no_f16_derived', {'ALQ_NO_F16_DERIVED': '1'}, None), ('bf16x3', {'ALQ_NO_F16X2': '1'}, None), ('fp32', {}, (4, 1)))
if os.environ.get('ACC_EXTRA_MASK'):      # study arm: more forward launches on fp16 pairs (ALQ_F16_DERIVED_MASK, layer bits)
    ARMS = ARMS[:1] + (('extra_mask_' + os.environ['ACC_EXTRA_MASK'], {'ALQ_F16_DERIVED_MASK': os.environ['ACC_EXTRA_MASK']}, None),) + ARMS[1:]
res = {}
for name, env, knob in ARMS:
    for k in ('ALQ_NO_F16_DERIVED', 'ALQ_NO_F16X2', 'ALQ_F16_DERIVED_MASK'):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=n)
    m.set_weights(pars)
    if knob:
        check(sess.lib.alq_debug_set(*knob))
    r = m.fisher_device(x, n, None, 1e-3, want=('p1', 'g0', 'g1'))
    torch.cuda.synchronize()
    if knob:
        check(sess.lib.alq_debug_set(knob[0], 0))
    res[name] = {k: r[k].cpu().numpy().copy() for k in ('p1', 'g0', 'g1')}
    m.close()
out = {'patches': n}
ref = res['fp32']
for name in [a_[0] for a_ in ARMS if a_[0] != 'fp32']:
    a = res[name]
    d = np.maximum(np.abs(a['g0'] - ref['g0']), np.abs(a['g1'] - ref['g1'])).max(axis=1)
    out[name] = {'over_2e-6': int((d > 2e-6).sum()), 'over_1e-5': int((d > 1e-5).sum()), 'over_1e-4': int((d > 1e-4).sum()),
                 'max_abs_dg': float(d.max()), 'max_abs_dp': float(np.abs(a['p1'] - ref['p1']).max()),
                 'rows_over_1e-4': np.nonzero(d > 1e-4)[0].tolist()}
    print(name, json.dumps(out[name]))
if arb:
    torch.set_num_threads(16)
    pars64 = {k: [v[0].astype(np.float64), v[1].astype(np.float64)] for k, v in pars.items()}
    om64 = OracleModel(ld, in_shape, pars64, skips=sk, dtype=torch.float64)
    xs = x.cpu().numpy()
    rows = out['default']['rows_over_1e-4']
    t0 = time.time()
    unexplained = []
    for i in rows:
        t1 = time.time()
        found = factored_ref.relu_flip_explains(om64, xs[i].reshape(in_shape).astype(np.float64),
                                                [(res[k]['g0'][i], res[k]['g1'][i]) for k in ('default', 'fp32')], 1e-3)
        print('patch %d: %s  (%.1f s)' % (i, found, time.time() - t1), flush=True)
        if any(f is None for f in found):
            unexplained.append(i)
    out['arbiter'] = {'rows': len(rows), 'seconds': time.time() - t0, 'unexplained': unexplained}
    print('arbiter', json.dumps(out['arbiter']))
print(json.dumps(out))
