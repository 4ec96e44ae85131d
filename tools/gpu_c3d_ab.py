#!/usr/bin/env python3
"""Plane-sweep engine (csrc/c3d.hip) against the two-slot engine (igemm4) on NET-C 32^3, GPU box only.

    python tools/gpu_c3d_ab.py [N_check] [N_time]

Two models in one process: ALQ_NO_C3D=1 (round-3 launches) and the default.  Prints the differences of p1, g0, g1 and the
median pass times of both, interleaved (cdna guide rule 24).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device  # noqa: E402
from nnal_amd._lib import check  # noqa: E402
from oracle import netspec  # noqa: E402


def main():
    n_chk = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n_time = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk, bias_std=0.05)
    NB = max(n_chk, n_time)
    models = {}
    for name, env in (('igemm4', {'ALQ_NO_C3D': '1'}), ('c3d', {}), ('c3d_2acc', {'ALQ_C3D_TWOACC': '1'})):
        for k in ('ALQ_NO_C3D', 'ALQ_C3D_TWOACC'):
            os.environ.pop(k, None)
        os.environ.update(env)
        m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=NB)
        m.set_weights(pars)
        models[name] = m
    for k in ('ALQ_NO_C3D', 'ALQ_C3D_TWOACC'):
        os.environ.pop(k, None)
    x = sess.empty((NB, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, NB, 32 ** 3, C.c_void_p(x.data_ptr())))
    info = lambda m, w: sess.lib.alq_model_engine_info(m._m, w)
    res = {}
    for name, m in models.items():
        r = m.fisher_device(x, n_chk, None, 1e-3, want=('p1', 'g0', 'g1', 'Asum'))
        torch.cuda.synchronize()
        res[name] = {k: v.detach().cpu().numpy().copy() for k, v in r.items() if v is not None}
        print('%-9s subnormals_ok=%d fwd_c3d=%d bwd_c3d=%d oneacc=%d' % (name, info(m, 0), info(m, 1), info(m, 2), info(m, 3)))
    ref = res['igemm4']
    for name in ('c3d', 'c3d_2acc'):
        r = res[name]
        for k in ('p1', 'g0', 'g1'):
            d = np.abs(r[k].astype(np.float64) - ref[k].astype(np.float64))
            rel = d / np.maximum(np.abs(ref[k].astype(np.float64)), 1e-30)
            print('%-9s %-3s max abs %.3e  median rel %.3e  p99 rel %.3e  max rel %.3e' % (name, k, d.max(), np.median(rel), np.percentile(rel, 99), rel.max()))
        d = np.abs(r['Asum'] - ref['Asum'])
        print('%-9s Asum max rel %.3e' % (name, (d / np.abs(ref['Asum']).max()).max()))
    times = {k: [] for k in models}
    for rnd in range(6):
        for name, m in models.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.fisher_device(x, n_time, None, 1e-3, want=('p1', 'Asum'))
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) * 1e3)
    for name in models:
        t = np.array(times[name][1:])
        print('%-9s median %.3f ms per %d-patch pass (min %.3f) -> %.0f patches/s' % (name, np.median(t), n_time, t.min(), n_time / np.median(t) * 1e3))


if __name__ == '__main__':
    main()
