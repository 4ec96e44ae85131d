#!/usr/bin/env python3
"""Throughput of the reference-literal patch net (NET-B = NN.create_PW1, 32 slices as channels) on the GPU box.

    python tools/gpu_netb.py [batch]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    sess = device.DeviceSession(0)
    ld = netspec.net_b()
    in_shape = (32, 32, 32)
    pars = netspec.he_init(ld, in_shape, seed=13)
    model = device.DeviceModel(sess, ld, in_shape, (), max_batch=N)
    model.set_weights(pars)
    nb = 8
    x = torch.randn((N * nb, 32 * 32 * 32), dtype=torch.float32, device=sess.device)
    ts = []
    for rnd in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fisher_device(x, N * nb, None, 1e-5, want=('p1', 'Asum'))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / nb)
    t = np.median(ts[1:])
    print('NET-B (create_PW1, 42.05 M parameters, [N,32,32,32]): %.3f ms / %d-patch pass -> %.0f patches/s Fisher-scored' %
          (t * 1e3, N, N / t))
    sess.prof_reset()
    sess.prof_enable(True)
    model.fisher_device(x, N * nb, None, 1e-5, want=('p1', 'Asum'))
    sess.prof_enable(False)
    for k, v in sess.prof_read().items():
        if v['launches']:
            print('   %-12s %8.3f ms per pass  %6d launches  %7.2f TF' % (k, v['ms'] / nb, v['launches'] / nb, v['flops'] / max(v['ms'], 1e-9) / 1e9))
    model.close()


if __name__ == '__main__':
    main()
