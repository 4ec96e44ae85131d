#!/usr/bin/env python3
"""Layer-by-layer comparison of the HIP path with the oracle (manual diagnostic, GPU box only):

    python tools/gpu_diag.py [neta|netb|netc2d|netc|all] > gpurun_out/diag.log
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# every activation and cotangent is read back below: keep the fc head as its own kernels (a fused head leaves the last
# conv's output and its cotangent unmaterialised in a Fisher pass and alq_model_debug_copy refuses them)
os.environ.setdefault('ALQ_NO_FC_BITS', '1')

import torch  # noqa: E402

import nnal_amd  # noqa: E402,F401
from nnal_amd import device  # noqa: E402
from oracle import netspec  # noqa: E402
from oracle.model import OracleModel  # noqa: E402
from tests import factored_ref  # noqa: E402


def cases():
    c = {}
    c['neta'] = (netspec.net_a(), (32, 32, 1), (), 4)
    c['neta6'] = (netspec.net_a(), (9, 9, 6), (), 5)
    c['netb'] = (netspec.net_b_small(), (25, 25, 2), (), 3)
    c['netb32'] = (netspec.net_b_small(width=64), (32, 32, 32), (), 2)
    lc, sk = netspec.net_c_2d()
    c['netc2d'] = (lc, (16, 16, 1), sk, 4)
    lc, sk = netspec.net_c()
    c['netc8'] = (lc, (8, 8, 8, 1), sk, 3)
    c['netc'] = (lc, (32, 32, 32, 1), sk, 2)
    return c


def err(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    if a.shape != b.shape:
        return 'SHAPE %s vs %s' % (a.shape, b.shape)
    d = np.abs(a - b)
    return 'max|d|=%.3e  rel=%.3e  (ref max %.3e)' % (d.max(), d.max() / (np.abs(b).max() + 1e-30), np.abs(b).max())


def run(name, ld, in_shape, skips, n, sess, wseed=5, xseed=3, max_batch=None):
    print('=' * 100)
    print('CASE', name, in_shape, 'n =', n)
    pars = netspec.he_init(ld, in_shape, seed=wseed, skips=skips, bias_std=0.05)
    om = OracleModel(ld, in_shape, pars, skips=skips)
    x = np.random.RandomState(xseed).randn(n, *in_shape).astype(np.float32)
    det = {}
    p, S, sizes = factored_ref.factored_unit_scores(om, x, det)
    t0 = time.time()
    dm = device.DeviceModel(sess, ld, in_shape, skips, max_batch=max_batch or max(n, 4))
    dm.set_weights(pars)
    print('  device model built in %.2fs' % (time.time() - t0))
    res = dm.fisher(x, None, 1e-5)
    sess.synchronize()
    print('  p1      ', err(res['p1'], p[1]))
    nd = len(in_shape) - 1
    for i, lname in enumerate(om.names):
        d = dm.debug_tensor(i, 0, n)
        ref = det['out'][i]
        if ref.ndim == 2:      # [F, N] -> [N, F]
            ref = ref.T
        print('  act  %-6s' % lname, err(d, ref))
    pi = 0
    for i, lname in enumerate(om.names):
        if dm.layers[i]['type'] == 2:
            continue
        ref = det['delta'][pi]
        if det['types'][pi] == 'fc':
            ref = ref.T
        print('  delta %-6s' % lname, err(dm.debug_tensor(i, 1, n), ref))
        print('  asum  %-6s' % lname, err(dm.debug_tensor(i, 2, n), det['asum'][pi]))
        print('  dsum  %-6s' % lname, err(dm.debug_tensor(i, 3, n), det['dsum'][pi]))
        pi += 1
    Sd = dm.debug_tensor(0, 4, n).reshape(n, -1)
    for t in range(S.shape[1]):
        print('  S[%d]     ' % t, err(Sd[:, t], S[:, t]), ' dev', Sd[:, t][:2], 'ref', S[:, t][:2])
    g0, g1, A = factored_ref.fisher_from_unit(p[1], S, sizes, 1e-5)
    print('  g0      ', err(res['g0'], g0))
    print('  g1      ', err(res['g1'], g1))
    print('  A       ', err(res['A'], A))
    print('  Asum    ', err(res['Asum'], A.sum(0)))
    dm.close()


def main():
    which = sys.argv[1:] or ['all']
    sess = device.DeviceSession(0)
    print(torch.cuda.get_device_name(0))
    allc = cases()
    if 'netc_golden' in which:
        ld, in_shape, skips, n = allc['netc']
        for mb in (4, 16):
            run('netc_golden_mb%d' % mb, ld, in_shape, skips, 3, sess, wseed=14, xseed=1004, max_batch=mb)
        run('netc_seed5_n3_mb16', ld, in_shape, skips, 3, sess, max_batch=16)
    for name, (ld, in_shape, skips, n) in allc.items():
        if 'all' in which or name in which:
            try:
                run(name, ld, in_shape, skips, n, sess)
            except Exception as e:   # keep going: one log for all cases
                import traceback
                traceback.print_exc(file=sys.stdout)
                print('CASE', name, 'FAILED:', e)
            sys.stdout.flush()


if __name__ == '__main__':
    main()
