"""BASELINE.json configs[2] at its FULL size on the GPU box - 100,000 synthetic 32^3 patches, NET-C, the bench's weights - checked through
size-independent properties of the scored path (no oracle can walk 100k patches of 866 MFLOP each in a test):

  * checksum of checksums: the reported pool sum of A_i == the fp64 sum of the stored A_i; tr A_i == the stored diagonals;
  * algebra the reference's formulas imply (PW_NNAL.py:770-814): A_i = (1 - p) g0 g0^T + p g1 g1^T + lambda I rebuilt from the stored
    p1, g0, g1; g0 and g1 are p1 and -p0 times ONE vector (p0 + p1 = 1 to fp32 rounding); A_i symmetric with diagonal >= lambda;
  * the uncertainty filter (PW_NNAL.py:671-681): the selected indices == a stable argsort of |p1 - .5| over the whole pool, bit-exact;
  * idempotence: a second scoring pass returns the same bits;
  * cut invariance / sharding: the pool scored as 4 contiguous shards gives the same per-patch bits, shard sums that add
    up to the pool sum, and a top-B merge equal to the whole pool's selection - what the N-GPU path relies on;
  * forward-only posteriors (the entropy filter's pass) == the Fisher pass's p1 bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import netspec  # noqa: E402

N_POOL = 100000
TOPB = 4096
LAMBDA = 1e-3


@pytest.fixture(scope='module')
def scored():
    import nnal_amd  # noqa: F401
    from nnal_amd import device, pool_shard
    from nnal_amd._lib import check
    sess = device.default_session()
    torch = sess.torch
    ld, sk = netspec.net_c()
    in_shape = (32, 32, 32, 1)
    model = device.DeviceModel(sess, ld, in_shape, sk, max_batch=2047)
    model.set_weights(netspec.he_init(ld, in_shape, seed=14, skips=sk))
    x = sess.empty((N_POOL, 32 ** 3), torch.float32)
    check(sess.lib.alq_synth_patches(sess.ctx, 1004, 0, N_POOL, 32 ** 3, C.c_void_p(x.data_ptr())))
    out = pool_shard.score_pool(model, sess, x, N_POOL, TOPB, LAMBDA)
    torch.cuda.synchronize()
    yield sess, model, x, out
    model.close()
    del x
    torch.cuda.empty_cache()


def test_checksums_and_the_algebra_of_the_stored_outputs(scored):
    sess, model, x, out = scored
    torch = sess.torch
    A, g0, g1, p1, tr = out['A'], out['g0'], out['g1'], out['p1'].double(), out['trace']
    assert A.shape == (N_POOL, 8, 8) and A.dtype == torch.float64 and bool(torch.isfinite(A).all())
    # pool sum: the library's fixed-order sum against torch's own fp64 reduction of the stored matrices
    S = A.sum(dim=0)
    assert float((out['Asum'] - S).abs().max()) <= 1e-9 * float(S.abs().max())
    # trace and symmetry
    assert bool((tr == torch.diagonal(A, dim1=1, dim2=2).sum(dim=1)).all()) or \
        float((tr - torch.diagonal(A, dim1=1, dim2=2).sum(dim=1)).abs().max()) <= 1e-15 * float(tr.abs().max())
    assert bool((A == A.transpose(1, 2)).all())
    assert float(torch.diagonal(A, dim1=1, dim2=2).min()) >= LAMBDA * (1 - 1e-12)
    # A_i from the stored p1, g0, g1 with the reference's saturation branches
    lo, hi = p1 < 1e-6, p1 > 1 - 1e-6
    p = torch.where(lo, torch.zeros_like(p1), torch.where(hi, torch.ones_like(p1), p1))
    R = (1 - p)[:, None, None] * g0[:, :, None] * g0[:, None, :] + p[:, None, None] * g1[:, :, None] * g1[:, None, :] \
        + LAMBDA * torch.eye(8, dtype=torch.float64, device=A.device)[None]
    assert float((A - R).abs().max()) <= 1e-12 * float(R.abs().max())
    # g0 = p1 g and g1 = -(1 - p1) g for one vector g: (1 - p1) g0 + p1 g1 = 0 away from the saturated branches
    mid = ~(lo | hi)
    z = (1 - p1)[:, None] * g0 + p1[:, None] * g1
    scale = float(torch.maximum(g0.abs(), g1.abs()).max())
    # (to fp32 rounding: the two cotangents use the fp32 posteriors p1 and p0 of the softmax, whose sum is 1 to 6e-8 only)
    assert float(z[mid].abs().max()) <= 2e-7 * scale
    assert int(mid.sum()) > N_POOL // 2


def test_the_selected_indices_are_the_stable_argsort_of_the_whole_pool(scored):
    sess, model, x, out = scored
    torch = sess.torch
    keys = (out['p1'].double() - 0.5).abs()
    order = torch.sort(keys, stable=True).indices[:TOPB]
    assert bool((out['sel'] == order).all())                      # ties -> lower index first, like np.argsort(kind='stable')
    k = keys[out['sel']]
    assert bool((k[1:] >= k[:-1]).all())


def test_a_second_pass_and_four_shards_give_the_same_bits(scored):
    import nnal_amd  # noqa: F401
    from nnal_amd import pool_shard
    sess, model, x, out = scored
    torch = sess.torch
    again = pool_shard.score_pool(model, sess, x, N_POOL, TOPB, LAMBDA)
    for k in ('p1', 'H', 'g0', 'g1', 'A', 'trace', 'Asum', 'sel'):
        assert bool((again[k] == out[k]).all()), k
    # four contiguous shards of ragged sizes (the last passes of the shards are cut differently from the whole pool's)
    cuts = [0, 24999, 50001, 77777, N_POOL]
    Asum = torch.zeros((8, 8), dtype=torch.float64, device=out['A'].device)
    keys, gidx = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r = model.fisher_device(x[a:b], b - a, None, LAMBDA, want=('p1', 'g0', 'g1', 'A', 'Asum'))
        for k in ('p1', 'g0', 'g1', 'A'):
            assert bool((r[k] == out[k][a:b]).all()), (k, a, b)
        Asum += r['Asum']
        loc, kk = sess.uncertainty_filter(r['p1'], min(TOPB, b - a), with_keys=True)
        keys.append(kk.cpu().numpy())
        gidx.append((loc + a).cpu().numpy())
    assert float((Asum - out['Asum']).abs().max()) <= 1e-12 * float(out['Asum'].abs().max())
    merged = pool_shard._topk_merge(np.concatenate(keys), np.concatenate(gidx), TOPB)
    np.testing.assert_array_equal(np.asarray(merged), out['sel'].cpu().numpy())


def test_forward_only_posteriors_equal_the_fisher_pass(scored):
    sess, model, x, out = scored
    post, _, _ = model.forward_device(x[:8188], 8188)
    assert bool((post[1] == out['p1'][:8188]).all())
