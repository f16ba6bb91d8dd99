"""CPU tests of the oracle's own fast forms (test infrastructure checking test infrastructure): the vectorised float64
`gradient_diff_fast` against the literal loop nest `gradient_diff` (which follows fft_backproplib.cu:709-753 line by line), and the
whole-network step `net_step` against the per-pair pieces the one-step tests use."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import np_ref as R


@pytest.mark.parametrize("dM,dD,Nk,Nl,block", [(4, 3, 3, 3, 64), (16, 8, 5, 5, 7), (5, 7, 5, 5, 64), (3, 1, 5, 5, 2), (8, 6, 3, 5, 5), (2, 2, 7, 7, 1)])
def test_gradient_diff_fast_equals_the_literal_loop_nest(dM, dD, Nk, Nl, block):
    rng = np.random.default_rng(dM * 100 + dD)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nl)); f = rng.uniform(-1, 1, (dD, dM, Nk, Nl))
    b = rng.uniform(-1, 1, dM); p = rng.uniform(-1, 1, dD)
    lit = R.gradient_diff(c, f, b, p)
    fast = R.gradient_diff_fast(c, f, b, p, block=block)
    for a, g in zip(lit, fast):
        assert a.shape == g.shape
        assert np.abs(a - g).max() <= 1e-13 * max(1.0, np.abs(a).max())
    rows = [0, dM * dD - 1, (dM * dD) // 2]
    part = R.gradient_diff_fast(c, f, b, p, rows=rows)
    assert np.array_equal(part[0], fast[0].reshape(dM * dD, Nk, Nl)[rows])
    assert np.array_equal(part[1], np.transpose(fast[1], (1, 0, 2, 3)).reshape(dM * dD, Nk, Nl)[rows])
    if dD == 1:       # fft.cu:724 needs d1 != d AND m1 != m: a single input channel has no partner (SURVEY B-9)
        assert not lit[0].any() and not fast[0].any()


def test_gradient_diff_fast_propagates_the_division_by_zero():
    """two coinciding kernels / biases: the source divides by zero (fft.cu:724-746, SURVEY B-9); both forms give non-finite values
    in the same elements"""
    rng = np.random.default_rng(3)
    c = rng.uniform(-1, 1, (3, 3, 3, 3)); f = rng.uniform(-1, 1, (3, 3, 3, 3))
    c[2, 2] = c[0, 0]
    b = np.array([0.5, 0.25, 0.5]); p = rng.uniform(-1, 1, 3)
    lit = R.gradient_diff(c, f, b, p); fast = R.gradient_diff_fast(c, f, b, p)
    for a, g in zip(lit, fast):
        assert np.array_equal(np.isfinite(a), np.isfinite(g))
    assert not np.isfinite(lit[0][0, 0]).all() and not np.isfinite(lit[2][0])


@pytest.mark.parametrize("sym,maxdiff", [(0, 0), (1, 1), (0, 1)])
def test_net_step_is_the_per_pair_iteration_on_the_forward_spectra(sym, maxdiff):
    rng = np.random.default_rng(17)
    D, N, maps, Nk, s, B = 3, 32, [4, 6], 5, 2, 2
    ws, dD = [], D
    for dM in maps:
        c = rng.uniform(-1, 1, (dM, dD, Nk, Nk))
        f = np.transpose(c, (1, 0, 2, 3)).copy() if sym else rng.uniform(-1, 1, (dD, dM, Nk, Nk))
        ws.append((c, rng.uniform(-1, 1, dM), f, rng.uniform(-1, 1, dD))); dD = dM
    xs = np.floor(rng.uniform(0, 256, (B, D, N, N)))
    w1, m1, mse, recon = R.net_step(xs, ws, None, s, 0.2, maxdiff=maxdiff, sym=sym, fast_diff=False)
    L = len(maps)
    net_c = [w[0] for w in ws] + [w[2] for w in ws[::-1]]; net_b = [w[1] for w in ws] + [w[3] for w in ws[::-1]]
    sp = [R.autoenc_fft(x, net_c, net_b, [s] * L + [-s] * L) for x in xs]
    assert np.array_equal(recon, np.stack([q[0][-1] for q in sp]))
    z = lambda a: np.zeros_like(a)
    for l in range(L):
        c, b, f, p = ws[l]
        Xs = [q[2][2 * l + 1] for q in sp]; Os = [q[2][4 * L - 1 - 2 * l] for q in sp]
        if not sym:
            r = R.batch_train_iter(Xs, Xs, Os, sp[0][1][l], sp[0][1][2 * L - 1 - l], c, f, b, p, (z(c), z(f), z(b), z(p)), 0.02, maxdiff)
            for a, k in zip(w1[l], ("c", "b", "f", "p")):
                assert np.allclose(a, r[k], rtol=0, atol=1e-14)       # (0.1 * 0.2 in net_step vs the literal 0.02 here: one ulp of the rate)
            assert abs(mse[l] - r["mse"]) <= 1e-12 * r["mse"]
        else:
            dck, dfk, db, dp = R.batch_grad(Xs, Xs, Os, sp[0][1][l], sp[0][1][2 * L - 1 - l], b, Nk, Nk)
            rc, rf, rb, rp = R.backprop_sym(c, f, b, p, dck, dfk, db, dp, z(c), z(f), z(b), z(p), 0.02, *R.gradient_diff(c, f, b, p))[:4]
            for a, r in zip(w1[l], (rc, rb, rf, rp)):
                assert np.allclose(a, r, rtol=0, atol=1e-14)
    # the second step carries the momentum
    w2, m2, _, _ = R.net_step(xs, w1, m1, s, 0.2, maxdiff=maxdiff, sym=sym, fast_diff=False)
    w2b, _, _, _ = R.net_step(xs, w1, None, s, 0.2, maxdiff=maxdiff, sym=sym, fast_diff=False)
    assert np.abs(w2[0][0] - w2b[0][0]).max() > 1e-5


def test_trajectory_fixture_is_reproduced_by_the_oracle():
    """tests/golden/traj300.npz (generator: tests/golden/make_traj.py) = np_ref.net_step on the seeded case: the first steps re-run here"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_traj as T
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj300.npz"))
    assert g["mse64"].shape == (T.CFG["steps"], len(T.CFG["maps"])) and g["mse32"].shape == g["mse64"].shape
    assert np.allclose(T.run(np.float64, 3), g["mse64"][:3], rtol=1e-12)
    assert np.allclose(T.run(np.float32, 2), g["mse32"][:2], rtol=1e-5)
