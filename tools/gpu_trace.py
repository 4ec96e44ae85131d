#!/usr/bin/env python3
"""A few NET-C Fisher passes under fixed alq_debug_set knobs, for rocprofv3 --kernel-trace (GPU box).

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 tools/gpu_trace.py "0=1" [batch]
    python tools/prof_seq.py out/t_kernel_trace.csv all
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device  # noqa: E402
from nnal_amd._lib import check  # noqa: E402
from oracle import netspec  # noqa: E402


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else ''
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=N)
    model.set_weights(pars)
    x = sess.empty((N * 5, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, N * 5, 32 ** 3, C.c_void_p(x.data_ptr())))
    for kv in filter(None, spec.split(',')):
        k, v = kv.split('=')
        check(sess.lib.alq_debug_set(int(k), int(v)))
    model.fisher_device(x, N * 5, None, 1e-3, want=('p1', 'Asum'))
    torch.cuda.synchronize()
    model.close()


if __name__ == '__main__':
    main()
