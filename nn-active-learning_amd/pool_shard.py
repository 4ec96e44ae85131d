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


def _comm_device():
    import torch
    dist = _dist()
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def allreduce_sum(mat):
    """Sum of a small float64 array over all ranks (the Fisher-matrix sum); identity at world 1."""
    rank, ws = world()
    mat = np.asarray(mat, dtype=np.float64)
    if ws == 1:
        return mat.copy()
    import torch
    t = torch.as_tensor(mat.copy()).to(_comm_device())
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    return t.cpu().numpy()


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


def score_pool(model, sess, local_patches, n_global, B, diag_load=1e-5, fisher_on='all'):
    """Scores this rank's shard of a pool of `n_global` patches.

    local_patches: device fp32 tensor [n_local, ...] = patches shard_bounds(n_global, R, rank).
    Returns dict: 'sel' global top-B most-uncertain positions (same on all ranks), 'Asum' the
    all-reduced sum of A_i over the pool (fisher_on='all') and the local per-patch outputs."""
    torch = sess.torch
    rank, ws = world()
    a, b = shard_bounds(n_global, ws, rank)
    n_local = b - a
    assert int(local_patches.shape[0]) == n_local
    out = model.fisher_device(local_patches, n_local, None, diag_load, want=('p1', 'trace', 'Asum'))
    from .PW_NNAL import device_uncertainty_filter
    Bl = min(B, n_local)
    if Bl > 0:
        loc = device_uncertainty_filter(sess, out['p1'], Bl)
        keys = (out['p1'][loc].double() - 0.5).abs().cpu().numpy()
        gidx = loc.cpu().numpy() + a
    else:
        keys, gidx = np.zeros(0), np.zeros(0, dtype=np.int64)
    sel = merge_topB(keys, gidx, min(B, n_global))
    Asum = allreduce_sum(out['Asum'].cpu().numpy())
    return dict(sel=sel, Asum=Asum, p1=out['p1'], trace=out['trace'], offset=a)
