#!/usr/bin/env python3
"""Run-to-run determinism of the scored path over the networks / shapes the tests use, with 1 .. 3 scoring pipelines sharing the
chip: every (net, shape, batch) is scored R times per pipeline count and compared bit for bit with its first single-pipeline run.
A kernel whose result depends on what else shares its compute units shows up here (csrc/e3d.hip did, round 6).

    python tools/determinism_probe.py [R]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402
from nnal_amd._lib import check  # noqa: E402


def cases():
    ld, sk = netspec.net_c()
    yield 'NET-C 32^3 batch 33', ld, sk, (32, 32, 32, 1), 33, 320
    yield 'NET-C 32^3 batch 61', ld, sk, (32, 32, 32, 1), 61, 427
    yield 'NET-C 16^3 batch 40', ld, sk, (16, 16, 16, 1), 40, 500
    yield 'NET-C 8^3 batch 100', ld, sk, (8, 8, 8, 1), 100, 1000
    ld2, sk2 = netspec.net_c_2d()
    yield 'NET-C 2-D 32^2 batch 64', ld2, sk2, (32, 32, 1), 64, 600
    yield 'NET-A 32^2 batch 100', netspec.net_a(), (), (32, 32, 1), 100, 1000
    yield 'NET-B 25^2 x 2 batch 96', netspec.net_b(), (), (25, 25, 2), 96, 700
    yield 'NET-B small 25^2 x 2 batch 50', netspec.net_b_small(), (), (25, 25, 2), 50, 500


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sess = device.DeviceSession(0)
    keys = ('p1', 'g0', 'g1', 'A')
    total_bad = 0
    for name, ld, sk, in_shape, B, n in cases():
        t0 = time.time()
        m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=B)
        m.set_weights(netspec.he_init(ld, in_shape, seed=14, skips=sk, bias_std=0.05))
        elems = int(np.prod(in_shape))
        x = sess.empty((n, elems), torch.float32)
        check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, elems, C.c_void_p(x.data_ptr())))
        m.lanes = 1
        o = m.fisher_device(x, n, None, 1e-3, want=keys)
        ref = {k: o[k].cpu().numpy() for k in keys}
        for lanes in (1, 2, 3):
            m.lanes = lanes
            bad = 0
            for r in range(R):
                o = m.fisher_device(x, n, None, 1e-3, want=keys)
                cur = {k: o[k].cpu().numpy() for k in keys}
                diff = [k for k in keys if not np.array_equal(cur[k], ref[k])]
                if diff:
                    bad += 1
                    d = np.abs(cur['g0'] - ref['g0'])
                    if bad <= 3:
                        print('   %s, %d pipelines, run %d: %s differ; patches %s layers %s max |dg0| %.3e' % (
                            name, lanes, r, diff, np.nonzero(d.max(axis=1) > 0)[0][:8].tolist(), np.nonzero(d.max(axis=0) > 0)[0].tolist(), d.max()), flush=True)
            print('%-32s %d pipelines: %d of %d runs differ' % (name, lanes, bad, R), flush=True)
            total_bad += bad
        # forward-only passes
        post0 = m.forward_device(x, n)[0].cpu().numpy()
        badf = sum(int(not np.array_equal(m.forward_device(x, n)[0].cpu().numpy(), post0)) for _ in range(R))
        print('%-32s forward-only: %d of %d runs differ   (%.1f s)' % (name, badf, R, time.time() - t0), flush=True)
        total_bad += badf
        m.close()
        del x
    print('TOTAL differing runs: %d' % total_bad, flush=True)
    return 1 if total_bad else 0


if __name__ == '__main__':
    sys.exit(main())
