#!/usr/bin/env python3
"""Diagnostic for the fused fc-head epilogue (igemm4 FCF): per (tile, wave) logit partials of 16 NET-C 32^3 patches
(8 + their copies: tiles of patches 0-7 are written back inside the tick loop, 8-15 after it), three passes, dumped
to an .npz for off-line comparison between builds of libalq (ALQ_LIB selects the build).

    ALQ_LIB=libalq_pk1.so python tests/gpu_fcf_diag.py gpurun_out/fcf_pk1.npz
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402
from nnal_amd._lib import check  # noqa: E402


def main():
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    x = sess.empty((16, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, 8, 32 ** 3, C.c_void_p(x.data_ptr())))
    x[8:] = x[:8]
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=16)
    m.set_weights(pars)
    parts, p1 = [], []
    for _ in range(int(os.environ.get('FCF_PASSES', '20'))):
        r = m.fisher_device(x, 16, None, 1e-3, want=('p1',))
        buf = sess.empty((16 * 4096,), torch.float32)
        e = C.c_int64()
        check(sess.lib.alq_model_debug_copy(m._m, 0, 5, 16, C.c_void_p(buf.data_ptr()), C.byref(e)))
        parts.append(buf[:e.value].cpu().numpy().reshape(16, -1))
        p1.append(r['p1'].cpu().numpy())
    parts, p1 = np.stack(parts), np.stack(p1)
    np.savez(sys.argv[1], parts=parts, p1=p1)
    same_runs = all(np.array_equal(parts[0], parts[i]) for i in range(1, len(parts)))
    same_halves = np.array_equal(parts[:, :8], parts[:, 8:])
    print('%s: partials %s, runs identical: %s, inside == after: %s, max |inside - after| = %.3e' %
          (os.environ.get('ALQ_LIB', 'libalq.so'), parts.shape, same_runs, same_halves, np.abs(parts[:, :8] - parts[:, 8:]).max()))
    m.close()


if __name__ == '__main__':
    main()
