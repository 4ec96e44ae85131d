#!/usr/bin/env python3
"""Same-process A/B timing of the NET-C Fisher pass under alq_debug_set knobs (GPU box only).

    python tools/gpu_ab.py "2=0" "2=1" ...      each argument = comma list of key=value knobs

Variants are interleaved round-robin (cdna guide rule 24): per variant the median ms per
256-patch pass and the per-class HIP-event shares are printed.
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
    variants = sys.argv[1:] or ['', '2=1']
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    N = int(os.environ.get('AB_BATCH', '256'))
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=N)
    model.set_weights(pars)
    nb = 8
    x = sess.empty((N * nb, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, N * nb, 32 ** 3, C.c_void_p(x.data_ptr())))

    def setk(spec, on):
        for kv in filter(None, spec.split(',')):
            k, v = kv.split('=')
            check(sess.lib.alq_debug_set(int(k), int(v) if on else 0))

    times = {v: [] for v in variants}
    for rnd in range(7):
        for v in variants:
            setk(v, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.fisher_device(x, N * nb, None, 1e-3, want=('p1', 'Asum'))
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) * 1e3 / nb)
            setk(v, False)
    for v in variants:
        t = np.array(times[v][1:])
        print('variant %-12r median %.3f ms / %d-patch pass  (min %.3f)  -> %.0f patches/s' %
              (v, np.median(t), N, t.min(), N / np.median(t) * 1e3))


if __name__ == '__main__':
    main()
