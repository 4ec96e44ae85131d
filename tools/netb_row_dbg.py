#!/usr/bin/env python3
"""Shipped engines against the exact-fp32 engine, tensor by tensor, for one patch of the NET-B (width 256, seed 17) test configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nnal_amd  # noqa: E402,F401
from nnal_amd import device, netspec  # noqa: E402
from nnal_amd._lib import check  # noqa: E402


def main():
    row = int(sys.argv[1]) if len(sys.argv) > 1 else 162
    sess = device.DeviceSession(0)
    shape = (32, 32, 32)
    n = 203
    ld = netspec.net_b_small(width=256)
    pars = netspec.he_init(ld, shape, seed=17, bias_std=0.02)
    x = sess.to_device(np.random.RandomState(22).randn(n, int(np.prod(shape))).astype(np.float32), torch.float32)
    m = device.DeviceModel(sess, ld, shape, (), max_batch=n)
    m.set_weights(pars)
    nl = len(m.layers)

    def dump():
        r = m.fisher_device(x, n, None, 1e-3, want=('p1', 'g0', 'g1'))
        torch.cuda.synchronize()
        out = {'g0': r['g0'].cpu().numpy()[row], 'p1': r['p1'].cpu().numpy()[row]}
        for li in range(nl):
            for what in (0, 1, 2, 3):
                try:
                    t = m.debug_tensor(li, what, n)
                    out[(li, what)] = t.reshape(n, -1)[row].copy()
                except Exception as e:
                    pass
        try:
            out['S'] = m.debug_tensor(0, 4, n).reshape(n, -1)[row].copy()
        except Exception:
            pass
        return out
    a = dump()
    check(sess.lib.alq_debug_set(4, 1))
    b = dump()
    check(sess.lib.alq_debug_set(4, 0))
    print('layers:', [(i, l['type'] if isinstance(l, dict) and 'type' in l else str(l)[:40]) for i, l in enumerate(m.layers)])
    print('g0 shipped', a['g0']); print('g0 exact  ', b['g0']); print('p1', a['p1'], b['p1'])
    names = {0: 'activation', 1: 'cotangent', 2: 'asum', 3: 'dsum'}
    for k in sorted([k for k in a if isinstance(k, tuple)]):
        if k in b and a[k].shape == b[k].shape:
            d = np.abs(a[k] - b[k])
            sc = max(np.abs(b[k]).max(), 1e-30)
            print('layer %2d %-10s n %7d  max|d| %.3e  scale %.3e  rel %.2e  n(d > 1e-5 scale) %d  first idx %s' % (
                k[0], names[k[1]], a[k].size, d.max(), sc, d.max() / sc, int((d > 1e-5 * sc).sum()), np.nonzero(d > 1e-5 * sc)[0][:6].tolist()))
        else:
            print('layer %2d %-10s only in %s' % (k[0], names[k[1]], 'shipped' if k in a else 'exact'))
    k = (4, 1)
    if k in a and k in b:
        d = np.abs(a[k] - b[k])
        for idx in np.nonzero(d > 1e-5 * np.abs(b[k]).max())[0][:4]:
            vox, ch = divmod(int(idx), 96)
            y, xx = divmod(vox, 16)
            y0, x0 = y & ~1, xx & ~1
            print('cotangent element', idx, 'voxel', (y, xx), 'channel', ch, 'shipped', a[k][idx], 'exact', b[k][idx])
            for dy in (0, 1):
                for dx in (0, 1):
                    j = ((y0 + dy) * 16 + x0 + dx) * 96 + ch
                    print('    window (%d, %d): activation shipped %.9g exact %.9g   cotangent shipped %.6g exact %.6g' % (y0 + dy, x0 + dx, a[(4, 0)][j], b[(4, 0)][j], a[k][j], b[k][j]))
    if 'S' in a and 'S' in b:
        print('S shipped', a['S']); print('S exact  ', b['S'])
    m.close()


if __name__ == '__main__':
    main()
