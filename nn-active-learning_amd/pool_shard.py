"""Multi-GPU scoring of one unlabelled pool: one process per GPU, contiguous block shards, no
data-path collective; the only exchanges are (1) the top-B merge of (|p-.5|, global index)
candidates and (2) one all-reduce(sum) of the L x L Fisher sum, over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The partition is the "concatenated sets" model of patch_utils.global2local_inds
(patch_utils.py:855-864): rank r owns global positions [r*ceil(n/R), ...), global = offset + local.
Neither collective exists in the reference (single process); see SURVEY.md §8(e)."""
import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def world():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n, world_size, rank):
    """Contiguous block [a, b) of rank `rank`; blocks of ceil(n/R), the last ones may be short/empty."""
    per = -(-int(n) // int(world_size))
    a = min(int(n), rank * per)
    return a, min(int(n), a + per)


def work_block(n):
    """[a, b): this rank's contiguous share of n independent work items whose inputs every rank holds (the voxels of
    replicated volumes): the same block rule as shard_bounds.  World 1: everything."""
    rank, ws = world()
    return shard_bounds(n, ws, rank)


def _comm_device():
    import torch
    dist = _dist()
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def attach_comm(sess):
    """Gives the session's libalq context its own RCCL communicator over the ranks of the process group
    (alq_comm_unique_id on rank 0 -> broadcast of the 128-byte id through torch.distributed -> alq_comm_init on
    every rank), so that the Fisher-sum all-reduce is the C ABI's alq_allreduce_sum on the library's stream.
    Returns True when the communicator exists afterwards.  Needs a GPU session; world 1 works (RCCL accepts a
    one-rank communicator), which is how the GPU test exercises it."""
    if getattr(sess, 'comm_world', 0) > 0:
        return True
    rank, ws = world()
    # vote 1 - BEFORE the collective init: can every rank reach RCCL at all (alq_comm_unique_id is the cheapest call that
    # needs the library)?  ncclCommInitRank blocks until every rank has joined, so a rank that cannot join must be
    # known before any rank enters it.
    err, my_uid = None, None
    try:
        my_uid = sess.comm_unique_id()
    except Exception as e:
        err = e
    if ws > 1 and max_over_ranks(0.0 if err is None else 1.0) > 0:
        raise RuntimeError('RCCL is not usable on %s' % ('this rank: %s' % (err,) if err is not None else 'another rank'))
    if err is not None:
        raise err
    uid = [my_uid if rank == 0 else None]
    if ws > 1:
        _dist().broadcast_object_list(uid, src=0)
    try:
        sess.comm_init(uid[0], rank, ws)
    except Exception as e:
        err = e
    # vote 2 - all or nothing: a rank whose init failed would fall back to torch.distributed while the others wait in the
    # library's all-reduce; the ranks that did get a communicator give it back, so a later attempt starts clean
    if ws > 1 and max_over_ranks(0.0 if err is None else 1.0) > 0:
        if err is None:
            sess.comm_destroy()
        sess.comm_world = 0
        raise RuntimeError('alq_comm_init failed on %s' % ('this rank: %s' % (err,) if err is not None else 'another rank'))
    if err is not None:
        raise err
    return True


def allreduce_sum(mat, sess=None):
    """Sum of a small float64 array over all ranks (the Fisher-matrix sum); identity at world 1.
    With a session that holds an RCCL communicator (attach_comm) the reduction is alq_allreduce_sum on the device;
    otherwise torch.distributed (gloo in the CPU tests)."""
    rank, ws = world()
    if sess is not None and getattr(sess, 'comm_world', 0) == ws:      # the context's communicator spans this process group
        torch = sess.torch
        if isinstance(mat, torch.Tensor):
            t = mat.to(dtype=torch.float64).contiguous().clone()
        else:
            t = sess.to_device(np.asarray(mat, dtype=np.float64), torch.float64)
        return sess.allreduce_sum_(t).cpu().numpy()
    if hasattr(mat, 'cpu'):
        mat = mat.cpu().numpy()
    mat = np.asarray(mat, dtype=np.float64)
    if ws == 1:
        return mat.copy()
    import torch
    t = torch.as_tensor(mat.copy()).to(_comm_device())
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    return t.cpu().numpy()


def allgather_rows(total, positions, rows, sess=None):
    """Every rank contributes `rows[i]` (float64) for the global row `positions[i]` of a [total, ...] array; all
    ranks receive the assembled array.  Realised as ONE all-reduce(sum) of an array that is zero outside a rank's
    own rows: x + 0.0 is exact, each row has exactly one owner, so the result is bit-identical to the owner's
    values (the all-gather of a ragged partition without a size exchange)."""
    rows = np.asarray(rows, dtype=np.float64)
    full = np.zeros((int(total),) + rows.shape[1:], dtype=np.float64)
    if len(positions):
        full[np.asarray(positions, dtype=np.int64)] = rows
    rank, ws = world()
    if ws == 1:
        return full
    return allreduce_sum(full, sess)


def _topk_merge(keys, gidx, B):
    """alq_topk_merge (include/alq.h): ascending key, ties -> lower global index, padding (index < 0) dropped."""
    import ctypes as C
    from ._lib import lib, check
    k = np.ascontiguousarray(keys, dtype=np.float64)
    g = np.ascontiguousarray(gidx, dtype=np.int64)
    out = np.empty(max(int(B), 1), dtype=np.int64)
    n_out = C.c_int64(0)
    check(lib().alq_topk_merge(k.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), len(k), int(B),
                               out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    return out[:n_out.value].copy()


def merge_topB(local_keys, local_global_idx, B):
    """Every rank passes its local candidates (ascending or not) as (key, GLOBAL index); returns the
    global top-B index list, ascending key, ties -> lower global index, identical on all ranks.
    Each rank needs to contribute at most its own best B."""
    rank, ws = world()
    k = np.asarray(local_keys, dtype=np.float64)
    g = np.asarray(local_global_idx, dtype=np.int64)
    if len(k) > B:
        o = np.lexsort((g, k))[:B]
        k, g = k[o], g[o]
    if ws > 1:
        import torch
        dist = _dist()
        dev = _comm_device()
        # fixed-size exchange: B (key, index) pairs per rank, padded with +inf / -1
        kk = np.full(B, np.inf)
        gg = np.full(B, -1, dtype=np.int64)
        kk[:len(k)] = k
        gg[:len(g)] = g
        tk = torch.as_tensor(kk).to(dev)
        tg = torch.as_tensor(gg).to(dev)
        lk = [torch.empty_like(tk) for _ in range(ws)]
        lg = [torch.empty_like(tg) for _ in range(ws)]
        dist.all_gather(lk, tk)
        dist.all_gather(lg, tg)
        k = torch.cat(lk).cpu().numpy()
        g = torch.cat(lg).cpu().numpy()
    return _topk_merge(k, g, B)


def merge_topB_device(sess, keys, gidx, B, n_global=None):
    """merge_topB on device tensors, nothing staged through the host: every rank passes its (at most B) best candidates as a
    float64 key tensor in ascending order (ties -> lower index) and their GLOBAL indices; fixed-size all-gather of the padded
    (key, index) vectors over the process group (RCCL under "nccl"), then the device top-B of the concatenation
    (alq_topk_uncertain: ascending key, ties -> lower POSITION, and position order IS global-index order among equal keys
    because ranks own ascending index blocks and each rank's list is already tie-ordered).  Returns the global indices as
    an int64 device tensor [min(B, candidates in total)], identical on every rank.

    n_global (the pool size; ranks own the blocks of shard_bounds): the number of real candidates in the gathered vector is
    then known on the host - sum over the ranks of min(B, block size) - and nothing in here waits for the device.  Without
    it the count is read back from the gathered indices (one host synchronisation per call)."""
    torch = sess.torch
    rank, ws = world()
    nl = int(keys.numel())
    if nl > B:
        keys, gidx, nl = keys[:B], gidx[:B], B
    dist = _dist()
    # a one-rank "nccl" group still takes the collective path: that is how the one-GPU box exercises the RCCL all-gather of
    # device tensors this function issues on a real node
    if ws > 1 or (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'):
        # padding: the largest key BIT PATTERN there is (the merge compares keys as bit patterns: numbers ascending, NaNs behind
        # them) - a real candidate, even one with a NaN key, never sorts behind a padding slot
        kk = torch.full((B,), 0x7fffffffffffffff, dtype=torch.int64, device=keys.device).view(torch.float64)
        gg = torch.full((B,), -1, dtype=torch.int64, device=keys.device)
        kk[:nl] = keys
        gg[:nl] = gidx
        if dist.get_backend() != 'nccl':                     # gloo rehearsal: the exchange runs on host tensors
            kk, gg = kk.cpu(), gg.cpu()
        K = torch.empty((ws * B,), dtype=torch.float64, device=kk.device)
        G = torch.empty((ws * B,), dtype=torch.int64, device=gg.device)
        dist.all_gather_into_tensor(K, kk)
        dist.all_gather_into_tensor(G, gg)
        K, G = K.to(keys.device), G.to(keys.device)
        if B <= 0:
            valid = 0
        elif n_global is not None:
            valid = sum(min(int(B), b - a) for a, b in (shard_bounds(n_global, ws, r) for r in range(ws)))
        else:
            valid = int((G >= 0).sum().item())               # padding must not be selected (n_global < B only)
    else:
        K, G, valid = keys.contiguous(), gidx.contiguous(), nl
    take = min(int(B), valid)
    if take == 0:
        return torch.empty((0,), dtype=torch.int64, device=keys.device)
    return G.index_select(0, sess.topk_smallest(K, take))


def allreduce_sum_device(t, sess):
    """In-place sum of a float64 device tensor over the ranks, result left on the device (no host copy): the context's RCCL
    communicator when it has one (alq_allreduce_sum), else torch.distributed on the tensor itself (nccl) or through the
    host (gloo rehearsal).  Identity at world 1."""
    rank, ws = world()
    if getattr(sess, 'comm_world', 0) == ws:          # (also a one-rank communicator: the GPU test's RCCL path)
        return sess.allreduce_sum_(t)
    if ws == 1:
        return t
    dist = _dist()
    if dist.get_backend() == 'nccl':
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    t.copy_(h)
    return t


def max_over_ranks(value):
    rank, ws = world()
    if ws == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64).to(_comm_device())
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return float(t.item())


def barrier():
    rank, ws = world()
    if ws > 1:
        _dist().barrier()


def score_pool(model, sess, local_patches, n_global, B, diag_load=1e-5,
               want=('p1', 'H', 'g0', 'g1', 'A', 'trace', 'Asum')):
    """Scores this rank's shard of a pool of `n_global` patches ("Fisher-scored", SURVEY.md 8d: per patch p1,
    |p1-.5| (through the top-B keys), H, g0, g1, A_i, tr A_i, and the pool sum of A_i).

    local_patches: device fp32 tensor [n_local, ...] = patches shard_bounds(n_global, R, rank).
    Returns dict of DEVICE tensors: 'sel' global top-B most-uncertain positions (int64, same on all ranks), 'Asum' the
    all-reduced sum of A_i over the pool, and the local per-patch outputs."""
    torch = sess.torch
    rank, ws = world()
    a, b = shard_bounds(n_global, ws, rank)
    n_local = b - a
    assert int(local_patches.shape[0]) == n_local
    out = model.fisher_device(local_patches, n_local, None, diag_load, want=want)
    Bl = min(B, n_local)
    if Bl > 0:
        loc, keys = sess.uncertainty_filter(out['p1'], Bl, with_keys=True)
        gidx = loc + a
    else:
        keys = sess.empty((0,), torch.float64)
        gidx = sess.empty((0,), torch.int64)
    # both exchanges stay on the device: nothing in this function copies to the host or synchronises (the number of real
    # candidates of the merge follows from n_global on the host)
    sel = merge_topB_device(sess, keys, gidx, min(B, n_global), n_global=n_global)
    Asum = allreduce_sum_device(out['Asum'], sess) if out.get('Asum') is not None else None
    out.update(sel=sel, Asum=Asum, offset=a)
    return out


# ------------------------------------------------------------------------------------------ rank launcher
def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(argv, n, env=None, timeout=None):
    """Starts `n` rank processes of `argv` (a command list) on this node - RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment, one process per GPU - relays rank 0's stdout and every
    rank's stderr, and returns (exit code, rank 0's stdout text).  The exit code is non-zero when ANY rank failed;
    the others are then terminated (a rank that died leaves its peers blocked in a collective).

    The caller must not have initialised the GPU: children are fresh processes (never an exec of a process that
    touched HIP).  This is what `python bench.py --gpus N` does from a bare shell."""
    import os
    import subprocess
    import sys
    import time
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import tempfile
    procs = []
    out0_file = tempfile.TemporaryFile(mode='w+')        # a file, not a pipe: nothing blocks however much rank 0 prints
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e, stdout=out0_file if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    t0 = time.time()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            c = procs[r].poll()
            if c is None:
                continue
            pending.discard(r)
            if c != 0 and rc == 0:
                rc = c
                print('[spawn_ranks] rank %d exited with code %d; stopping the others' % (r, c), file=sys.stderr, flush=True)
                for q in pending:
                    procs[q].terminate()
        if pending:
            if timeout is not None and time.time() - t0 > timeout:
                rc = rc or 124
                for q in pending:
                    procs[q].kill()
                timeout = None
            time.sleep(0.05)
    out0_file.seek(0)
    out0 = out0_file.read()
    out0_file.close()
    return rc, out0
