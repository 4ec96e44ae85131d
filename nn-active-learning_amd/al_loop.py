"""Query rounds of the reference's active-learning loop at the patch-tensor level.

Control flow of `PW_AL.Experiment_MultiImg.run_method` (PW_AL.py:690-898) between two fine-tunes, with
the volume I/O and the fine-tune itself left out (both outside the hot path): per round

    1. posteriors of every patch still in the pool               (PW_NNAL.bin_uncertainty_filter_multimg -> batch_eval)
    2. the B most uncertain ones, |p - .5| ascending             (PW_NNAL.py:729-730)
    3. their conditional Fisher matrices A_i                     (PW_NNAL.gen_A_matrices, diag_load 1e-3: PW_NNAL.py:578)
    4. the query distribution over the B candidates              (NNAL_tools.SDP_query_distribution, lambda = 0: PW_NNAL.py:596-604)
    5. k draws from it with the global NumPy RNG                 (NNAL_tools.sample_query_dstr: PW_NNAL.py:617-620)
    6. the drawn patches leave the pool                          (PW_AL.py:870-882)

Everything up to the A_i runs on the device through the C ABI; steps 4-5 are host NumPy like the reference's
(the SDP solver is this build's own - a log-barrier Newton method on the A-optimal design the SDP states: cvxopt is
absent, parity unpinned - see NNAL_tools.SDP_query_distribution).
"""
import time

import numpy as np

from . import NNAL_tools
from .PW_NNAL import device_uncertainty_filter


def run_rounds(model, sess, pool, rounds, B, k, diag_load=1e-3, seed=15, chunk=8192):
    """pool: device fp32 tensor [n, ...] of normalised patches.  Returns a list with one dict per round:
    'queries' (positions into `pool`, sorted), 'candidates' (the B filtered positions), 'posts' of the candidates,
    'A' [B, L, L], 'q' the query distribution, 'sdp' solver report and wall times per stage."""
    torch = sess.torch
    n = int(pool.shape[0])
    flat = pool.reshape(n, -1)
    remaining = np.arange(n, dtype=np.int64)
    out = []
    for r in range(rounds):
        t0 = time.perf_counter()
        nr = len(remaining)
        rem_dev = sess.to_device(remaining, torch.int64)
        p1 = sess.empty((nr,), torch.float32)
        for a in range(0, nr, chunk):                      # gather a chunk of the remaining patches, score it
            b = min(nr, a + chunk)
            x = flat.index_select(0, rem_dev[a:b])
            post, _, _ = model.forward_device(x, b - a)
            p1[a:b] = post[1]
        Bq = min(int(B), nr)
        cand_local = device_uncertainty_filter(sess, p1, Bq)           # positions into `remaining`
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        xc = flat.index_select(0, rem_dev.index_select(0, cand_local))
        res = model.fisher_device(xc, Bq, p1.index_select(0, cand_local), diag_load, want=('A',))
        A = res['A'].cpu().numpy()
        t2 = time.perf_counter()
        soln = NNAL_tools.SDP_query_distribution(A, 0, None, k)
        q = np.array(soln['x'][:Bq], dtype=np.float64)
        np.random.seed(seed + r)
        draws = NNAL_tools.sample_query_dstr(q, k, replacement=True)
        t3 = time.perf_counter()
        cand = remaining[cand_local.cpu().numpy()]
        queries = np.sort(cand[draws])
        remaining = np.setdiff1d(remaining, queries, assume_unique=True)
        out.append(dict(queries=queries, candidates=cand, posts=p1.index_select(0, cand_local).cpu().numpy(), A=A, q=q,
                        sdp={kk: soln[kk] for kk in ('status', 'primal objective', 'gap', 'iterations')},
                        seconds=dict(filter=t1 - t0, fisher=t2 - t1, sdp_and_sampling=t3 - t2), pool_left=len(remaining)))
    return out


def main():
    """python -m nnal_amd.al_loop [pool] [rounds]: NET-C, synthetic 32^3 pool (seed 1005), weights seed 15."""
    import ctypes as C
    import sys
    from . import device
    from ._lib import check
    from . import netspec
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    sess = device.DeviceSession(0)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=512)
    model.set_weights(netspec.he_init(ld, in_shape, seed=15, skips=sk))
    pool = sess.empty((n, 32 ** 3), sess.torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1005, 0, n, 32 ** 3, C.c_void_p(pool.data_ptr())))
    for r, rd in enumerate(run_rounds(model, sess, pool, rounds, 4096, 100)):
        print('round %d: %d queries, pool left %d, filter %.2f s, fisher %.2f s, sdp+sampling %.2f s (%s, %d iterations)' %
              (r, len(rd['queries']), rd['pool_left'], rd['seconds']['filter'], rd['seconds']['fisher'],
               rd['seconds']['sdp_and_sampling'], rd['sdp']['status'], rd['sdp']['iterations']))
    model.close()


if __name__ == '__main__':
    main()
