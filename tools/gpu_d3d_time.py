#!/usr/bin/env python3
"""Kernel times of a few full-batch Fisher passes from the library's own HIP events, by engine class (GPU box); env switches apply."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402
from nnal_amd._lib import check  # noqa: E402
n = 2047
sess = device.DeviceSession(0)
ld, sk = netspec.net_c()
in_shape = (32, 32, 32, 1)
pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
x = sess.empty((n, 32 ** 3), torch.float32)
check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=n)
m.set_weights(pars)
for _ in range(3):
    m.fisher_device(x, n, None, 1e-3)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    m.fisher_device(x, n, None, 1e-3)
torch.cuda.synchronize()
print('%.3f ms per pass' % ((time.perf_counter() - t0) / 10 * 1e3))
