#!/usr/bin/env python3
"""Which decision explains row 162 of the NET-B (width 256, seed 17) test configuration, and at what key?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec, ref64  # noqa: E402
from nnal_amd._lib import check  # noqa: E402


def main():
    sess = device.DeviceSession(0)
    shape = (32, 32, 32)
    n = 203
    ld = netspec.net_b_small(width=256)
    pars = netspec.he_init(ld, shape, seed=17, bias_std=0.02)
    x = sess.to_device(np.random.RandomState(22).randn(n, int(np.prod(shape))).astype(np.float32), torch.float32)
    for name, env in (('default', {}), ('fc fwd bf16x3', {'ALQ_NO_FC_F16_FWD': '1'}), ('no wide2d rule', {'ALQ_NO_WIDE2D_RULE': '1'}),
                      ('all bf16x3', {'ALQ_NO_F16X2': '1'})):
        os.environ.update(env)
        m = device.DeviceModel(sess, ld, shape, (), max_batch=n)
        m.set_weights(pars)
        for k in env:
            os.environ.pop(k)
        r = m.fisher_device(x, n, None, 1e-3, want=('p1', 'g0', 'g1'))
        eng = {name: (r['g0'].cpu().numpy(), r['g1'].cpu().numpy())}
        r64 = ref64.Ref64(m, max_samples=64)
        for eps in (4e-6, 2e-5, 1e-4):
            rep, base, found = r64.engine_report(x, np.arange(n), eng, eps=eps)
            print(name, 'eps', eps, {k: rep[name][k] for k in ('over_2e-6', 'over_1e-4', 'max_abs_dg', 'flips_needed', 'unexplained_rows')}, rep['_fragility'], flush=True)
            if rep[name]['flips_needed']['unexplained'] == 0:
                f = found[0][162]
                print('   row 162 flips:', f)
                break
        m.close()


if __name__ == '__main__':
    main()
