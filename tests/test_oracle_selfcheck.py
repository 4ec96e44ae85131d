"""Pins the part of the oracle the reference cannot pin (the TensorFlow op semantics): torch
calls vs loop-level numpy twins, fp64 finite differences of the log-posterior gradients, and
the factored identity the HIP kernels rely on.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import alpath, netspec, tfops
from oracle.model import OracleModel, OracleSession
from tests import factored_ref


@pytest.mark.parametrize('shape,k,s', [((2, 7, 6, 3), (3, 3), None), ((1, 5, 5, 2), (5, 5), None),
                                       ((1, 4, 5, 3, 2), (3, 3, 3), None), ((1, 7, 6, 2), (3, 3), (2, 2))])
def test_conv_same_vs_naive(shape, k, s):
    rs = np.random.RandomState(0)
    x = rs.randn(*shape)
    W = rs.randn(*(k + (shape[-1], 4)))
    b = rs.randn(4)
    y = tfops.conv_same(torch.tensor(x), torch.tensor(W), torch.tensor(b), s).numpy()
    np.testing.assert_allclose(y, tfops.naive_conv_same(x, W, b, s), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('shape,w', [((2, 7, 5, 3), (2, 2)), ((1, 5, 4, 3, 2), (2, 2, 2)), ((1, 25, 25, 1), (2, 2))])
def test_max_pool_same_vs_naive(shape, w):
    x = np.random.RandomState(1).randn(*shape)
    y = tfops.max_pool_same(torch.tensor(x), w, w).numpy()
    np.testing.assert_array_equal(y, tfops.naive_max_pool_same(x, w, w))
    assert y.shape[1] == -(-shape[1] // 2)      # 25 -> 13: the odd unit of padding goes to the end


@pytest.mark.parametrize('shape,k,s', [((2, 4, 3, 3), (3, 3), (2, 2)), ((1, 3, 2, 4, 2), (3, 3, 3), (2, 2, 2)),
                                       ((1, 4, 4, 2), (2, 2), (2, 2))])
def test_conv_transpose_same_vs_naive(shape, k, s):
    rs = np.random.RandomState(2)
    x = rs.randn(*shape)
    W = rs.randn(*(k + (5, shape[-1])))
    b = rs.randn(5)
    y = tfops.conv_transpose_same(torch.tensor(x), torch.tensor(W), torch.tensor(b), s).numpy()
    assert y.shape[1] == shape[1] * s[0]
    np.testing.assert_allclose(y, tfops.naive_conv_transpose_same(x, W, b, s), rtol=1e-12, atol=1e-12)


def test_conv_transpose_is_gradient_of_same_conv():
    """tf.nn.conv*_transpose is DEFINED as the input-gradient of the SAME conv (no flip)."""
    rs = np.random.RandomState(3)
    x = torch.tensor(rs.randn(1, 3, 4, 2))                     # small map
    W = torch.tensor(rs.randn(3, 3, 5, 2))                     # [k,k,out,in] of the transpose
    big = torch.zeros(1, 6, 8, 5, dtype=torch.float64, requires_grad=True)
    y = tfops.conv_same(big, W, torch.zeros(2, dtype=torch.float64), (2, 2))   # filter [k,k,in=5,out=2]
    (y * x).sum().backward()
    np.testing.assert_allclose(tfops.conv_transpose_same(x, W, torch.zeros(5, dtype=torch.float64), (2, 2)).numpy(),
                               big.grad.numpy(), rtol=1e-12, atol=1e-12)


def test_flatten_order():
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float64).reshape(2, 3, 4, 5)    # N,H,W,C
    f = tfops.flatten_tf(x).numpy()
    for (h, w, c) in [(0, 0, 0), (2, 1, 3), (1, 3, 4)]:
        assert f[c * (4 * 3) + w * 3 + h, 1] == x[1, h, w, c]                  # f = c*(W*H) + w*H + h


def _models():
    out = []
    ld = netspec.net_a()
    out.append(('neta', ld, (12, 12, 2), ()))
    out.append(('netb', netspec.net_b_small(width=16), (9, 9, 2), ()))
    lc, sk = netspec.net_c_2d()
    out.append(('netc2d', lc, (8, 8, 1), sk))
    lc, sk = netspec.net_c()
    out.append(('netc', lc, (4, 4, 4, 1), sk))
    return out


@pytest.mark.parametrize('name,ld,in_shape,skips', _models())
def test_grad_finite_difference_fp64(name, ld, in_shape, skips):
    pars = netspec.he_init(ld, in_shape, seed=5, skips=skips, bias_std=0.1, dtype=np.float64)
    x = np.random.RandomState(6).randn(1, *in_shape)
    model = OracleModel(ld, in_shape, pars, skips=skips, dtype=torch.float64)
    for j in (0, 1):
        grads = model.grad_log_post(j, x)
        rs = np.random.RandomState(7)
        for t, (lname, (W, b)) in enumerate(pars.items()):
            for arr, gr in ((W, grads[2 * t]), (b, grads[2 * t + 1])):
                idx = tuple(rs.randint(0, s) for s in arr.shape)
                eps = 1e-6
                vals = []
                for sgn in (+1, -1):
                    p2 = {k: [v[0].copy(), v[1].copy()] for k, v in pars.items()}
                    tgt = p2[lname][0] if arr is W else p2[lname][1]
                    tgt[idx] += sgn * eps
                    m2 = OracleModel(ld, in_shape, p2, skips=skips, dtype=torch.float64)
                    vals.append(np.log(m2.forward(x)['posteriors'][j, 0]))
                fd = (vals[0] - vals[1]) / (2 * eps)
                assert abs(fd - gr[idx]) <= 1e-6 * max(1., abs(fd)), (name, lname, idx)


@pytest.mark.parametrize('name,ld,in_shape,skips', _models())
def test_factored_identity_fp64(name, ld, in_shape, skips):
    """One unit-cotangent backward + channel-sum/box-sum reductions == summed full gradients."""
    pars = netspec.he_init(ld, in_shape, seed=8, skips=skips, bias_std=0.1, dtype=np.float64)
    model = OracleModel(ld, in_shape, pars, skips=skips, dtype=torch.float64)
    sess = OracleSession(model)
    x = np.random.RandomState(9).randn(3, *in_shape)
    p, S, sizes = factored_ref.factored_unit_scores(model, x)
    g0f, g1f, _ = factored_ref.fisher_from_unit(p[1], S, sizes, 1e-5)
    for i in range(3):
        g0, g1 = alpath.shrunk_grads(model, sess, x[i])
        np.testing.assert_allclose(g0f[i], g0, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(g1f[i], g1, rtol=1e-9, atol=1e-14)


def test_fp64_arbiter_finds_a_max_pool_near_tie():
    """tests/factored_ref.relu_flip_explains (the fp64 arbiter of the GPU parity tests) also considers max-pool windows whose two
    largest inputs lie within rounding of each other: deciding such a tie the other way leaves every value unchanged and moves
    the window's whole cotangent to another voxel - the layer scores below the pool jump like they do for a ReLU flip.
    Here: the scores of an evaluation with the most fragile window of a small NET-C patch decided the other way are handed
    to the arbiter, which must name exactly that window."""
    import torch
    from oracle import netspec
    from oracle.model import OracleModel
    from tests import factored_ref
    ld, sk = netspec.net_c()
    in_shape = (8, 8, 8, 1)
    pars = netspec.he_init(ld, in_shape, seed=3, skips=sk)
    pars64 = {k: [v[0].astype(np.float64), v[1].astype(np.float64)] for k, v in pars.items()}
    om = OracleModel(ld, in_shape, pars64, skips=sk, dtype=torch.float64)
    x = np.random.RandomState(1).randn(*in_shape)
    det = {}
    p, S, sizes = factored_ref.factored_unit_scores(om, x[None], det)
    xin = det['pool_in']['pool1']
    v = xin.reshape(1, 4, 2, 4, 2, 4, 2, 8).transpose(0, 1, 3, 5, 7, 2, 4, 6).reshape(-1, 8)
    ii = np.arange(xin.size).reshape(xin.shape).reshape(1, 4, 2, 4, 2, 4, 2, 8).transpose(0, 1, 3, 5, 7, 2, 4, 6).reshape(-1, 8)
    srt = np.argsort(-v, axis=1)
    top, sec = v[np.arange(len(v)), srt[:, 0]], v[np.arange(len(v)), srt[:, 1]]
    ok = np.nonzero(top > 0)[0]
    w = ok[np.argmin((top - sec)[ok])]
    bump = np.zeros(xin.size)
    bump[ii[w, srt[w, 1]]] = 2 * (top[w] - sec[w]) + 1e-12
    pf, Sf, _ = factored_ref.factored_unit_scores(om, x[None], None, flips={'pool:pool1': bump.reshape(xin.shape)})
    g0, g1, _ = factored_ref.fisher_from_unit(p[1], S, sizes, 1e-3)
    f0, f1, _ = factored_ref.fisher_from_unit(pf[1], Sf, sizes, 1e-3)
    assert np.abs(pf - p).max() < 1e-3 and np.abs(f0 - g0).max() > 1e-7       # values move by (almost) nothing, scores do
    rms = float(np.sqrt(np.mean(xin ** 2)))
    found = factored_ref.relu_flip_explains(om, x, [(f0[0], f1[0]), (g0[0], g1[0])], 1e-3, atol=1e-9, rtol=1e-7,
                                            eps=1.01 * (top[w] - sec[w]) / rms + 1e-15, max_units=200, max_flips=1)
    assert found[1] == ()
    assert found[0] == (('pool:pool1', int(ii[w, srt[w, 1]])),), found


def test_fp64_arbiter_rejects_wrong_scores_and_non_fragile_flips():
    """The arbiter must not explain away a WRONG score (round-4 verdict, weak 2).  NET-C at 8^3 in fp64, fragility window widened
    to eps = 1e-3 of the layer rms so that it has candidates to combine: (1) the exact scores with one entry moved by 1e-5 (five
    times the absolute bar) are rejected; (2) the scores of an evaluation in which a ReLU unit FAR from zero (|pre| >= 0.5 rms) is
    inverted are rejected at the default window and at the widened one; (3) the control - the most fragile ReLU unit inverted -
    is accepted and named."""
    ld, sk = netspec.net_c()
    in_shape = (8, 8, 8, 1)
    pars = netspec.he_init(ld, in_shape, seed=5, skips=sk, bias_std=0.05)
    pars64 = {k: [v[0].astype(np.float64), v[1].astype(np.float64)] for k, v in pars.items()}
    om = OracleModel(ld, in_shape, pars64, skips=sk, dtype=torch.float64)
    x = np.random.RandomState(2).randn(*in_shape)
    det = {}
    p, S, sizes = factored_ref.factored_unit_scores(om, x[None], det)
    g0, g1, _ = factored_ref.fisher_from_unit(p[1], S, sizes, 1e-3)
    base = (g0[0].copy(), g1[0].copy())
    assert factored_ref.relu_flip_explains(om, x, [base], 1e-3) == [()]
    # (1) a perturbed score
    for layer in (0, 3):
        bad0 = base[0].copy()
        bad0[layer] += 1e-5
        assert abs(base[0][layer]) < 0.4            # 1e-5 is beyond 2e-6 + 2e-5 |g|
        for eps in (2e-5, 1e-3):
            assert factored_ref.relu_flip_explains(om, x, [(bad0, base[1])], 1e-3, eps=eps) == [None], (layer, eps)
    # (2) a unit far from zero, inverted
    relu_layers = [(nm, pre) for nm, pre, r in zip(det['names'], det['pre'], det['relu']) if r]
    name, pre = [(nm, pre) for nm, pre in relu_layers if nm == 'enc2'][0]
    rms = float(np.sqrt(np.mean(pre ** 2)))
    flat = np.abs(pre.reshape(-1)) / rms
    far = int(np.argmax(flat >= 0.5))
    fl = np.zeros(pre.size, bool)
    fl[far] = True
    pf, Sf, _ = factored_ref.factored_unit_scores(om, x[None], None, flips={name: fl.reshape(pre.shape)})
    f0, f1, _ = factored_ref.fisher_from_unit(pf[1], Sf, sizes, 1e-3)
    assert np.abs(f0[0] - base[0]).max() > 1e-5     # the wrong decision moves the scores well beyond the bars
    for eps in (2e-5, 1e-3):
        assert factored_ref.relu_flip_explains(om, x, [(f0[0], f1[0])], 1e-3, eps=eps) == [None], eps
    # (3) control: the most fragile unit of the patch
    best = min(((float(np.abs(pr.reshape(-1)).min() / np.sqrt(np.mean(pr ** 2))), nm, int(np.argmin(np.abs(pr.reshape(-1)))), pr.shape)
                for nm, pr in relu_layers), key=lambda t: t[0])
    fl = np.zeros(int(np.prod(best[3])), bool)
    fl[best[2]] = True
    pf, Sf, _ = factored_ref.factored_unit_scores(om, x[None], None, flips={best[1]: fl.reshape(best[3])})
    f0, f1, _ = factored_ref.fisher_from_unit(pf[1], Sf, sizes, 1e-3)
    found = factored_ref.relu_flip_explains(om, x, [(f0[0], f1[0])], 1e-3, atol=1e-9, rtol=1e-7, eps=1.01 * best[0] + 1e-15, max_units=50, max_flips=1)
    assert found == [((best[1], best[2]),)] or np.abs(f0[0] - base[0]).max() <= 1e-9, found
