#!/usr/bin/env python3
"""Run-to-run determinism probe under two pipelines: scores the same 320 patches R times (model batch B) and reports which
patches / layers differ from the first run.   python tools/race_probe.py [B] [R]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402
from nnal_amd._lib import check  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=B)
    m.set_weights(netspec.he_init(ld, in_shape, seed=14, skips=sk, bias_std=0.05))
    n = 320
    x = sess.empty((n, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
    keys = ('p1', 'g0', 'g1')
    ref = None
    bad = 0
    for r in range(R):
        o = m.fisher_device(x, n, None, 1e-3, want=keys)
        cur = {k: o[k].cpu().numpy() for k in keys}
        if ref is None:
            ref = cur
            continue
        d = np.abs(cur['g0'] - ref['g0'])
        if d.max() > 0 or np.abs(cur['p1'] - ref['p1']).max() > 0:
            bad += 1
            rows = np.nonzero(d.max(axis=1) > 0)[0]
            print('run %d: %d patches differ: %s; layers %s; max |dg0| %.3e (scale %.3e); dp1 %.3e' % (
                r, len(rows), rows[:12].tolist(), np.nonzero(d.max(axis=0) > 0)[0].tolist(), d.max(), np.abs(ref['g0']).max(),
                np.abs(cur['p1'] - ref['p1']).max()), flush=True)
    print('batch %d lanes %d env %s: %d of %d runs differ from the first' % (B, m.lanes, {k: v for k, v in os.environ.items() if k.startswith('ALQ_')}, bad, R - 1), flush=True)


if __name__ == '__main__':
    main()
