#!/usr/bin/env python3
"""Which elements of enc1's channel-sum field (layer 0, what = 3) differ between runs under two pipelines?"""
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
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=B)
    m.set_weights(netspec.he_init(ld, in_shape, seed=14, skips=sk, bias_std=0.05))
    n = 320
    x = sess.empty((n, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
    # the cut fisher_device will make: the main model's last pass
    step, starts = m.pass_cut(n)
    last_main = [a for k, a in enumerate(starts) if k % 2 == 0][-1]
    nlast = min(n, last_main + step) - last_main
    ref = None
    bad = 0
    for r in range(R):
        m.fisher_device(x, n, None, 1e-3, want=('g0',))
        torch.cuda.synchronize()
        d = m.debug_tensor(0, 3, nlast).reshape(nlast, 32, 32, 32)
        d2 = m.debug_tensor(2, 3, nlast).reshape(nlast, 16, 16, 16)
        if ref is None:
            ref, ref2 = d.copy(), d2.copy()
            continue
        w = np.argwhere(d != ref)
        w2 = np.argwhere(d2 != ref2)
        if len(w) or len(w2):
            bad += 1
            print('run %d: %d elements of enc1 dsum differ, %d of enc2 dsum' % (r, len(w), len(w2)))
            for q in w[:8]:
                print('   patch %d z %d y %d x %d: %r -> %r' % (last_main + q[0], q[1], q[2], q[3], float(ref[tuple(q)]), float(d[tuple(q)])))
    print('%d of %d runs differ (pass of %d patches at %d)' % (bad, R - 1, nlast, last_main), flush=True)


if __name__ == '__main__':
    main()
