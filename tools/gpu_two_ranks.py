#!/usr/bin/env python3
"""Functional rehearsal of the N > 1 paths on a ONE-GPU box: 1 rank vs 2 ranks that share cuda:0 (gloo exchanges; RCCL
refuses two ranks on one device).  The sharded bench step and the sharded config-5 loop (filter -> Fisher -> SDP ->
draws -> fine-tune) must give the same selections / queries / weights with 2 ranks as with 1, bit for bit.

    python tools/gpu_two_ranks.py gpurun_out/two_ranks"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nnal_amd  # noqa: E402,F401
from nnal_amd import pool_shard  # noqa: E402


def main():
    out = sys.argv[1]
    os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
    env = dict(os.environ, ALQ_DIST_BACKEND='gloo', ALQ_SAME_GPU='1', ALQ_LOOP_B='256', ALQ_LOOP_K='20')
    loop = [sys.executable, '-c', 'import sys; sys.path.insert(0, %r); import nnal_amd; from nnal_amd import al_loop; al_loop.main()' % ROOT,
            '6000', '3']
    e1 = dict(env, ALQ_LOOP_DUMP=out + '_w1')
    e1.pop('WORLD_SIZE', None)
    r = subprocess.run(loop, env=e1, capture_output=True, text=True)
    print(r.stdout[-600:], r.stderr[-300:] if r.returncode else '')
    assert r.returncode == 0
    rc, txt = pool_shard.spawn_ranks(loop, 2, env=dict(env, ALQ_LOOP_DUMP=out + '_w2'), timeout=600)
    print(txt[-600:])
    assert rc == 0, rc
    a = np.load(out + '_w1.rank0.npz')
    for rank in (0, 1):
        b = np.load(out + '_w2.rank%d.npz' % rank)
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (k, rank)
    print('sharded loop: 2 ranks == 1 rank, bit for bit (%d arrays: queries, candidates, posteriors, A, q of 3 rounds, final weights)' % len(a.files))
    # strong-scaling bench step over 2 ranks (same device): the JSON line must come out with n_gpus = 2
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--same-gpu', '--pool-global', '12000',
           '--batch', '1000', '--steps', '1', '--warmup', '1', '--no-cpu-baseline']
    e2 = dict(os.environ)
    e2.pop('WORLD_SIZE', None)
    r = subprocess.run(cmd, env=e2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['config']['pool_global'] == 12000 and line['value'] > 0
    print('bench parent -> 2 ranks: ok (%s)' % line['config']['parallelism'])


if __name__ == '__main__':
    main()
