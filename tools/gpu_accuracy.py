#!/usr/bin/env python3
"""Accuracy of the device Fisher scores against an fp64 evaluation of the same network (GPU box).

    python tools/gpu_accuracy.py [npatches]

Prints, for the bf16x3 GEMM path, the fp32-MFMA GEMM path (alq_debug_set(4,1)) and the fp32
torch-CPU oracle, the distribution of |g - g_fp64| / |g_fp64| over layers and patches: it shows how
far each fp32-level implementation sits from the exact value, ReLU-boundary flips included."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device  # noqa: E402
from nnal_amd._lib import check  # noqa: E402
from oracle import netspec  # noqa: E402
from oracle.model import OracleModel  # noqa: E402
from tests import factored_ref  # noqa: E402


def stats(name, g, ref):
    big = np.abs(ref) > 1e-6
    rel = np.abs(g - ref)[big] / np.abs(ref)[big]
    print('%-22s rel err: median %.2e  p90 %.2e  p99 %.2e  max %.2e   abs max %.2e' %
          (name, np.median(rel), np.percentile(rel, 90), np.percentile(rel, 99), rel.max(), np.abs(g - ref).max()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.set_num_threads(16)
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    pars = netspec.he_init(ld, in_shape, seed=14, skips=sk)
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=max(n, 4))
    model.set_weights(pars)
    x = sess.empty((n, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, n, 32 ** 3, C.c_void_p(x.data_ptr())))
    xs = x.cpu().numpy().reshape((n,) + in_shape)
    pars64 = {k: [v[0].astype(np.float64), v[1].astype(np.float64)] for k, v in pars.items()}
    om64 = OracleModel(ld, in_shape, pars64, skips=sk, dtype=torch.float64)
    p64, S64, sizes = factored_ref.factored_unit_scores(om64, xs.astype(np.float64))
    g64, _, _ = factored_ref.fisher_from_unit(p64[1], S64, sizes, 1e-3)
    om32 = OracleModel(ld, in_shape, pars, skips=sk)
    p32, S32, _ = factored_ref.factored_unit_scores(om32, xs)
    g32, _, _ = factored_ref.fisher_from_unit(p32[1], S32, sizes, 1e-3)
    stats('torch-CPU fp32 oracle', g32[:, :-1], g64[:, :-1])
    for name, knob in (('device bf16x3 MFMA', 0), ('device fp32 MFMA', 1)):
        check(sess.lib.alq_debug_set(4, knob))
        r = model.fisher_device(x, n, None, 1e-3, want=('g0', 'p1'))
        g = r['g0'].cpu().numpy()
        stats(name, g[:, :-1], g64[:, :-1])
        print('   p1 max abs err vs fp64: %.2e' % np.abs(r['p1'].cpu().numpy() - p64[1]).max())
    check(sess.lib.alq_debug_set(4, 0))


if __name__ == '__main__':
    main()
