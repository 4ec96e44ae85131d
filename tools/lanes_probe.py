#!/usr/bin/env python3
"""How many scoring pipelines pay?  L independent (session, model) pairs on L torch streams score passes of the 100k-patch pool
round-robin (the product's fisher_device does this with two).   python tools/lanes_probe.py [pool] [passes]"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    s0 = device.DeviceSession(0)
    x = s0.empty((n, 32 ** 3), torch.float32)
    check(s0.lib.alq_synth_patches(s0.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
    torch.cuda.synchronize()
    lanes = []
    for i in range(4):
        st = torch.cuda.Stream(s0.device)
        with torch.cuda.stream(st):
            s = device.DeviceSession(0)
        m = device.DeviceModel(s, ld, in_shape, sk, max_batch=2047)
        m.lanes = 1
        m.set_weights(pars)
        lanes.append((st, s, m))
    want = ('p1', 'H', 'g0', 'g1', 'A', 'trace', 'Asum')
    for L in (1, 2, 3, 4, 2, 3, 1):
        for st, s, m in lanes:
            check(s.lib.alq_ctx_use_side_stream(s.ctx, 1 if L == 1 else 0))
        P = -(-n // 2047)
        P = -(-P // L) * L
        step = -(-n // P)
        starts = list(range(0, n, step))
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cur = torch.cuda.current_stream()
            for st, s, m in lanes[:L]:
                st.wait_stream(cur)
            for k, a in enumerate(starts):
                st, s, m = lanes[k % L]
                b = min(n, a + step)
                with torch.cuda.stream(st):
                    m.fisher_device(x[a:b], b - a, None, 1e-3, want=want)
            for st, s, m in lanes[:L]:
                cur.wait_stream(st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print('lanes %d: %d passes of %d: %.1f k patches/s' % (L, len(starts), step, n / dt / 1e3), flush=True)


if __name__ == '__main__':
    main()
