#!/usr/bin/env python3
"""Times the conv_transpose launches of a NET-C Fisher pass with HIP events (classes igemm3_fwd: up2 forward when it is the only
bf16x3 ... ) - simply: wall time of N Fisher passes of one batch; compare libraries / env knobs in one gpurun call."""
import ctypes as C
import os
import sys
import time
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
m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=n)
m.set_weights(pars)
x = sess.empty((n, 32 ** 3), torch.float32)
check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
for _ in range(3):
    m.fisher_device(x, n, None, 1e-3, want=('p1',))
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 20
for _ in range(R):
    m.fisher_device(x, n, None, 1e-3, want=('p1',))
torch.cuda.synchronize()
print('%s %s: %.3f ms per 2047-patch Fisher pass' % (os.environ.get('ALQ_LIB', 'libalq.so'), os.environ.get('ALQ_T3D_WGS', ''), 1e3 * (time.perf_counter() - t0) / R))
