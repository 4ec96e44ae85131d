#!/usr/bin/env python3
"""Per-phase shader-clock shares of the pipelined GEMM kernel (diagnostic build libalq_stamps.so).

    ALQ_LIB=libalq_stamps.so python tools/gpu_stamps.py      (GPU box only)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('ALQ_LIB', 'libalq_stamps.so')

import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device  # noqa: E402
from nnal_amd._lib import check  # noqa: E402
from oracle import netspec  # noqa: E402

NAMES = ['A:stash', 'A:flush', 'A:locate+fetch', 'B:contract', 'wait after A', 'wait after B', 'tail', '-']
if os.environ.get('STAMP_V3'):
    NAMES = ['prologue', 'barrierA', 'stash', 'barrierB', 'epilogue', 'locate+fetch', 'mfma', 'tail']


def main():
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    N = int(os.environ.get("STAMP_BATCH", "256"))
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=N)
    model.set_weights(pars)
    x = sess.empty((N, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, N, 32 ** 3, C.c_void_p(x.data_ptr())))
    model.fisher_device(x, N, None, 1e-3)        # warm-up
    torch.cuda.synchronize()
    # stamp every GEMM launch separately: run the pass once per launch index is not possible from here, so
    # the buffer keeps the LAST pipelined launch of the pass; select it with ALQ_STAMP_ONLY (launch ordinal)
    buf = torch.zeros((4096 * 8,), dtype=torch.int64, device=sess.device)
    check(sess.lib.alq_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())))
    model.fisher_device(x, N, None, 1e-3)
    torch.cuda.synchronize()
    check(sess.lib.alq_debug_set_stamp_buffer(None))
    b = buf.cpu().numpy().reshape(-1, 8)
    b = b[b.sum(1) > 0]
    tot = b.sum(1)
    print('workgroups stamped: %d; ticks per WG: median %.0f' % (len(b), np.median(tot)))
    for i, n in enumerate(NAMES):
        print('  %-14s %6.1f %%   (median %9.0f ticks)' % (n, 100 * b[:, i].sum() / tot.sum(), np.median(b[:, i])))


if __name__ == '__main__':
    main()
