#!/usr/bin/env python3
"""Launch constants of the FORWARD-ONLY pass of NET-C at 32^3 (the entropy filter of the AL loop: alq_forward without sums) for
tools/gen_igemm4_fixed.py:   ALQ_DUMP_ARGS=1 python tools/dump_forward.py 2> gpurun_out/tunedump_forward.err     (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402


def main():
    sess = device.DeviceSession(0)
    torch = sess.torch
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    m = device.DeviceModel(sess, ld, in_shape, sk, max_batch=2000)
    m.set_weights(pars)
    x = sess.to_device(np.random.RandomState(3).randn(64, 32 ** 3).astype(np.float32), torch.float32)
    post, _, _ = m.forward_device(x, 64)
    torch.cuda.synchronize()
    print('forward-only pass done', float(post.sum()))
    m.close()


if __name__ == '__main__':
    main()
