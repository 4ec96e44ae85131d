#!/usr/bin/env python3
"""Scores a pool of P patches (default 100,000; NET-C 32^3, bench weights) R times under the default pipelines and compares every
output bit with the first run.     python tools/fullsize_soak.py [R] [P]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402
from nnal_amd._lib import check  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=2047)
    m.set_weights(netspec.he_init(ld, in_shape, seed=14, skips=sk))
    x = sess.empty((P, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, P, 32 ** 3, C.c_void_p(x.data_ptr())))
    keys = ('p1', 'g0', 'g1', 'A', 'Asum')
    ref, bad = None, 0
    for r in range(R):
        o = m.fisher_device(x, P, None, 1e-3, want=keys)
        torch.cuda.synchronize()
        if ref is None:
            ref = {k: o[k].clone() for k in keys}
            continue
        diff = [k for k in keys if not bool((o[k] == ref[k]).all())]
        if diff:
            bad += 1
            rows = torch.nonzero((o['g0'] != ref['g0']).any(dim=1)).flatten()[:8].tolist()
            print('run %d: %s differ; patches %s' % (r, diff, rows), flush=True)
    print('%d patches, %d pipelines: %d of %d runs differ from the first' % (P, m.lanes, bad, R - 1), flush=True)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
