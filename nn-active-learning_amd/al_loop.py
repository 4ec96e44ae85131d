"""Query rounds of the reference's active-learning loop at the patch-tensor level, on one GPU or
sharded over the ranks of a torch.distributed process group (one process per GPU).

Control flow of `PW_AL.Experiment_MultiImg.run_method` (PW_AL.py:690-898) between two fine-tunes: per round

    1. posteriors of every patch still in the pool               (PW_NNAL.bin_uncertainty_filter_multimg -> batch_eval)
    2. the B most uncertain ones, |p - .5| ascending             (PW_NNAL.py:729-730)
    3. their conditional Fisher matrices A_i                     (PW_NNAL.gen_A_matrices, diag_load 1e-3: PW_NNAL.py:578)
    4. the query distribution over the B candidates              (NNAL_tools.SDP_query_distribution: PW_NNAL.py:596-604)
    5. k draws from it with the global NumPy RNG                 (NNAL_tools.sample_query_dstr: PW_NNAL.py:617-620)
    6. the drawn patches leave the pool                          (PW_AL.py:870-882)
    7. the queries and the round's wall time go to `queries/<iter>` / `AL_running_times/dt_<iter>`   (PW_AL.py:862-885)
    8. fine-tune on everything labelled so far, `epochs` x random batches of `b`, then `curr_weights_<iter>`
                                                                 (PW_AL.finetune_multimg, PW_AL.py:890-898, :1091-1147)
   Steps 7-8 run when `state` / `finetune` are given; a run that finds `queries/` populated resumes behind the last
   complete iteration (PW_AL.py:724-735).

Sharding (no counterpart in the reference, which is one process): the pool is the concatenation of the ranks'
contiguous blocks (the "concatenated sets" model of patch_utils.global2local_inds, patch_utils.py:855-864;
pool_shard.shard_bounds) and a patch never moves: rank r scores the patches of ITS block that are still in the pool
(step 1), offers its best B to the top-B merge (step 2: pool_shard.merge_topB, identical result on every rank),
computes the A_i of the candidates it owns (step 3) and the B x L x L block is assembled on every rank by one
all-reduce of owner-filled rows (pool_shard.allgather_rows); steps 4-6 are then the same deterministic host code on
the same bits on every rank, so no further exchange is needed.  At world size 1 every exchange is the identity.

Everything up to the A_i runs on the device through the C ABI (row lists into the resident pool: alq_forward_rows /
alq_fisher_rows, no gathered copies); steps 4-5 are host NumPy like the reference's (the SDP solver is this build's
own - cvxopt is absent, parity unpinned - see NNAL_tools.SDP_query_distribution).
"""
import os
import time

import numpy as np

from . import NN, NNAL_tools, pool_shard


def _finetune(model, sess, x_train, y_train, epochs, b, keep_prob):
    """PW_AL.finetune_multimg's inner loops (PW_AL.py:1103-1146) on patches that are already tensors: `epochs` passes,
    NN.gen_batch_inds batches (global NumPy stream), one train_step per batch at keep_prob = model.dropout_rate."""
    n = int(x_train.shape[0])
    torch = sess.torch
    losses = []
    for _ in range(epochs):
        for batch in NN.gen_batch_inds(n, b):
            idx = sess.to_device(np.asarray(batch, dtype=np.int64), torch.int64)
            lab = y_train[np.asarray(batch)]
            hot = np.zeros((2, len(batch)))
            hot[0, lab == 0] = 1
            hot[1, lab == 1] = 1
            losses.append(model.train_on_batch(x_train.index_select(0, idx), hot, keep_prob=keep_prob))
    return losses


def run_rounds(model, sess, pool, rounds, B, k, diag_load=1e-3, seed=15, n_global=None, lambda_=0., after_round=None,
               labels=None, finetune=None, state=None):
    """pool: this rank's block of the pool, device fp32 tensor [n_local, ...] of normalised patches = global positions
    pool_shard.shard_bounds(n_global, world, rank) (n_global defaults to n_local: one process).  Returns a list with
    one dict per round, identical on every rank except 'seconds': 'queries' (GLOBAL positions, sorted), 'candidates'
    (the B filtered positions, most uncertain first), 'posts' of the candidates, 'A' [B, L, L], 'q' the query
    distribution, 'sdp' solver report, wall times per stage, 'pool_left' (global).

    labels [n_global] (0 / 1) + finetune = dict(epochs, b) (and the model's get_optimizer): after every round the model
    is fine-tuned on all patches labelled so far - on EVERY rank, on the same batches in the same order, so the
    replicas stay bit-identical without a weight exchange (the queried patches are gathered from their owners).
    state: a PW_AL.LoopState; rank 0 writes the files, every rank reads them to resume."""
    torch = sess.torch
    rank, ws = pool_shard.world()
    n_local = int(pool.shape[0])
    if n_global is None:
        n_global = n_local
    off, end = pool_shard.shard_bounds(n_global, ws, rank)
    if end - off != n_local:
        raise ValueError('rank %d holds %d patches, its block of a %d-patch pool over %d ranks has %d' %
                         (rank, n_local, n_global, ws, end - off))
    flat = pool.reshape(n_local, int(np.prod(pool.shape[1:])))      # (an empty shard has no -1 to infer)
    remaining = np.arange(n_local, dtype=np.int64)          # LOCAL positions of this rank's patches still in the pool
    left_global = int(n_global)
    out = []
    x_train = None                                          # [n_labelled, elems] on every rank
    trained = np.zeros(0, dtype=np.int64)                   # their global positions, in labelling order
    if finetune is not None and labels is None:
        raise ValueError('fine-tuning needs `labels`')

    def gather_patches(gpos):
        """The patches at global positions `gpos` on every rank (owners contribute, one all-reduce of owner-filled rows)."""
        own = np.nonzero((gpos >= off) & (gpos < end))[0]
        rows = flat.index_select(0, sess.to_device(gpos[own] - off, torch.int64)).cpu().numpy() if len(own) else \
            np.zeros((0, int(flat.shape[1])), np.float32)
        full = pool_shard.allgather_rows(len(gpos), own, rows, sess)
        return sess.to_device(full.astype(np.float32), torch.float32)

    first = 0
    if state is not None and state.iters_done() > 0:       # resume behind the last complete iteration (PW_AL.py:724-735)
        first = state.iters_done()
        done = state.load_queries()[:, 0]
        own = done[(done >= off) & (done < end)] - off
        remaining = np.setdiff1d(remaining, own, assume_unique=False)
        left_global -= len(done)
        trained = done.astype(np.int64)
        if finetune is not None:
            x_train = gather_patches(trained)
            if os.path.exists(state.weights_path(first)):
                model.load_weights(state.weights_path(first))
    for r in range(first, rounds):
        t0 = time.perf_counter()
        nr = len(remaining)
        rem_dev = sess.to_device(remaining, torch.int64)
        if nr > 0:
            post, _, _ = model.forward_device(flat, nr, rows=rem_dev)
            p1 = post[1].contiguous()
            Bl = min(int(B), nr)
            loc = sess.uncertainty_filter(p1, Bl)                           # positions into `remaining`
            p_loc = p1.index_select(0, loc)
            keys = (p_loc.double() - 0.5).abs().cpu().numpy()
            loc_h = loc.cpu().numpy()
            gidx = remaining[loc_h] + off
        else:
            keys, gidx, loc_h = np.zeros(0), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
            p_loc = sess.empty((0,), torch.float32)
        Bq = min(int(B), left_global)
        cand = pool_shard.merge_topB(keys, gidx, Bq)                        # global positions, same on every rank
        if hasattr(sess, 'synchronize'):
            sess.synchronize()
        t1 = time.perf_counter()
        mine = np.nonzero((cand >= off) & (cand < end))[0]                 # the candidates this rank owns
        if len(mine):
            # a candidate of ours is one of our local top-B: find it there to reuse its posterior
            order = np.argsort(gidx, kind='stable')
            at = order[np.searchsorted(gidx[order], cand[mine])]
            rows_dev = sess.to_device(cand[mine] - off, torch.int64)
            p_mine = p_loc.index_select(0, sess.to_device(at, torch.int64)).contiguous()
            res = model.fisher_device(flat, len(mine), p_mine, diag_load, want=('A',), rows=rows_dev)
            A_mine = res['A'].cpu().numpy()
            posts_mine = p_mine.cpu().numpy().astype(np.float64)
        else:
            A_mine = np.zeros((0, model.L, model.L))
            posts_mine = np.zeros(0)
        A = pool_shard.allgather_rows(len(cand), mine, A_mine, sess)
        posts = pool_shard.allgather_rows(len(cand), mine, posts_mine, sess).astype(np.float32)
        t2 = time.perf_counter()
        soln = NNAL_tools.SDP_query_distribution(A, lambda_, None, k)
        q = np.array(soln['x'][:len(cand)], dtype=np.float64)
        np.random.seed(seed + r)
        draws = NNAL_tools.sample_query_dstr(q, k, replacement=True)
        t3 = time.perf_counter()
        queries = np.sort(cand[draws])
        own = queries[(queries >= off) & (queries < end)] - off
        remaining = np.setdiff1d(remaining, own, assume_unique=True)
        left_global -= len(queries)
        rec = dict(queries=queries, candidates=cand, posts=posts, A=A, q=q,
                   sdp={kk: soln[kk] for kk in ('status', 'primal objective', 'gap', 'iterations')},
                   seconds=dict(filter=t1 - t0, fisher=t2 - t1, sdp_and_sampling=t3 - t2), pool_left=left_global)
        if state is not None and rank == 0:
            owner = np.minimum(queries // max(-(-int(n_global) // ws), 1), ws - 1)
            state.save_round(r, np.stack([queries, owner], axis=1), t3 - t0)          # dt = the query's wall time (:848-855)
        if finetune is not None:
            t4 = time.perf_counter()
            xq = gather_patches(queries)
            x_train = xq if x_train is None else torch.cat([x_train, xq])
            trained = np.concatenate([trained, queries])
            np.random.seed(seed + 7919 * (r + 1))           # same batches on every rank
            rec['finetune_loss'] = _finetune(model, sess, x_train, np.asarray(labels)[trained], finetune['epochs'], finetune['b'],
                                             finetune.get('keep_prob', model.dropout_rate))
            rec['seconds']['finetune'] = time.perf_counter() - t4
            if state is not None and rank == 0:
                model.save_weights(state.weights_path(r + 1))
        out.append(rec)
        if after_round is not None:
            after_round(r, queries)
        pool_shard.barrier()
    return out


def main():
    """python -m nnal_amd.al_loop [pool] [rounds] [state dir]: config 5 - NET-C, synthetic 32^3 pool (seed 1005), weights
    seed 15, per round entropy filter to B = 4096 -> Fisher -> SDP -> k = 100 draws -> fine-tune (SGD 1e-4, one epoch of
    batches of 50 over everything labelled so far; labels = sign of the patch's first 512 voxels' sum) -> weights saved.
    Under torch.distributed.run (WORLD_SIZE > 1) the pool is sharded over the ranks."""
    import ctypes as C
    import sys
    from . import device, PW_AL
    from ._lib import check
    from . import netspec
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    state = PW_AL.LoopState(sys.argv[3]) if len(sys.argv) > 3 else None
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('ALQ_DIST_BACKEND', 'nccl')       # gloo + ALQ_SAME_GPU=1: functional rehearsal on a one-GPU box
    if os.environ.get('ALQ_SAME_GPU'):
        local_rank = 0
    if ws > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=ws, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=ws)
    sess = device.DeviceSession(local_rank)
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=2000)
    model.set_weights(netspec.he_init(ld, in_shape, seed=15, skips=sk))
    model.get_optimizer(1e-4, [], 'SGD')
    a, b = pool_shard.shard_bounds(n, ws, rank)
    pool = sess.empty((b - a, 32 ** 3), sess.torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1005, a, b - a, 32 ** 3, C.c_void_p(pool.data_ptr())))
    lab_local = (pool[:, :512].sum(dim=1) > 0).cpu().numpy().astype(np.float64)
    labels = pool_shard.allgather_rows(n, np.arange(a, b), lab_local, sess).astype(np.int64)
    if ws > 1 and backend == 'nccl':
        try:        # a transport choice, not a compute fallback: torch.distributed's RCCL group does the same sums
            pool_shard.attach_comm(sess)
        except Exception as e:
            if rank == 0:
                print('al_loop: %s; the Fisher sums go through torch.distributed' % (e,), flush=True)
    dump = os.environ.get('ALQ_LOOP_DUMP')                      # rehearsals: every rank saves its rounds for comparison
    res = run_rounds(model, sess, pool, rounds, int(os.environ.get('ALQ_LOOP_B', '4096')), int(os.environ.get('ALQ_LOOP_K', '100')),
                     n_global=n, labels=labels, finetune=dict(epochs=1, b=50), state=state)
    if dump:
        np.savez('%s.rank%d.npz' % (dump, rank), **{'%s_%d' % (key, r): rd[key] for r, rd in enumerate(res)
                                                     for key in ('queries', 'candidates', 'posts', 'A', 'q')},
                 **{'w_' + nme.replace('/', '_'): wb[0] for nme, wb in model.var_dict.items()})
    for r, rd in enumerate(res):
        if rank == 0:
            print('round %d: %d queries, pool left %d, filter %.2f s, fisher %.2f s, sdp+sampling %.2f s (%s, %d iterations), '
                  'fine-tune %.2f s (%d steps, loss %.4f -> %.4f)' %
                  (r, len(rd['queries']), rd['pool_left'], rd['seconds']['filter'], rd['seconds']['fisher'],
                   rd['seconds']['sdp_and_sampling'], rd['sdp']['status'], rd['sdp']['iterations'], rd['seconds']['finetune'],
                   len(rd['finetune_loss']), rd['finetune_loss'][0], rd['finetune_loss'][-1]))
    model.close()
    if ws > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
