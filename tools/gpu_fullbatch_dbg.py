#!/usr/bin/env python3
"""Which patches of a 2000-patch batch differ between engines, and what fp64 says about the worst ones (GPU box)."""
import ctypes as C
import os
import sys
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
sess = device.DeviceSession(0)
ld, sk = netspec.net_c()
in_shape = (32, 32, 32, 1)
pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
x = sess.empty((n, 32 ** 3), torch.float32)
check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
res = {}
for name, env, knob in (('c3d', {}, None), ('igemm4', {'ALQ_NO_C3D': '1'}, None), ('bf16x3', {'ALQ_NO_F16X2': '1'}, None), ('fp32', {}, (4, 1))):
    for k in ('ALQ_NO_C3D', 'ALQ_NO_F16X2'):
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
names = list(res)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        d = np.maximum(np.abs(res[a]['g0'] - res[b]['g0']), np.abs(res[a]['g1'] - res[b]['g1']))
        bad = np.nonzero((d > 2e-6).any(axis=1))[0]
        print('%-7s vs %-7s: max |dg| %.3e, patches over 2e-6: %d %s; max |dp| %.2e' % (a, b, d.max(), len(bad), bad[:12].tolist(), np.abs(res[a]['p1'] - res[b]['p1']).max()))
d = np.maximum(np.abs(res['c3d']['g0'] - res['fp32']['g0']), np.abs(res['c3d']['g1'] - res['fp32']['g1']))
worst = np.argsort(-d.max(axis=1))[:4]
if len(sys.argv) > 2:
    worst = [int(v) for v in sys.argv[2].split(',')]
torch.set_num_threads(16)
pars64 = {k: [v[0].astype(np.float64), v[1].astype(np.float64)] for k, v in pars.items()}
om64 = OracleModel(ld, in_shape, pars64, skips=sk, dtype=torch.float64)
xs = x.cpu().numpy()
for i in worst:
    p64, S64, sizes = factored_ref.factored_unit_scores(om64, xs[i].reshape((1,) + in_shape).astype(np.float64))
    g64, h64, _ = factored_ref.fisher_from_unit(p64[1], S64, sizes, 1e-3)
    print('patch %d: p64 %.6f' % (i, p64[1][0]))
    for nme in names:
        print('   %-7s |g0 - g64| per layer %s' % (nme, np.array2string(np.abs(res[nme]['g0'][i] - g64[0]), precision=1)))
    found = factored_ref.relu_flip_explains(om64, xs[i].reshape(in_shape).astype(np.float64), [(res[nme]['g0'][i], res[nme]['g1'][i]) for nme in names], 1e-3)
    print('   flips:', found)
    if any(f is None for f in found):
        det = {}
        factored_ref.factored_unit_scores(om64, xs[i].reshape((1,) + in_shape).astype(np.float64), det)
        for name, pre, relu in zip(det['names'], det['pre'], det['relu']):
            if relu:
                rms = float(np.sqrt(np.mean(pre ** 2)))
                fl = np.abs(pre.reshape(-1)) / rms
                o = np.argsort(fl)[:4]
                print('     layer %-5s rms %.3e smallest |pre|/rms: %s at %s' % (name, rms, np.array2string(fl[o], precision=2), o.tolist()))
        for e2 in (1e-4, 5e-4):
            f2 = factored_ref.relu_flip_explains(om64, xs[i].reshape(in_shape).astype(np.float64), [(res[nme]['g0'][i], res[nme]['g1'][i]) for nme in names], 1e-3, eps=e2, max_units=16)
            print('     eps %.0e:' % e2, f2)
