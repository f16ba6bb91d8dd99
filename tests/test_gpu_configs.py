"""BASELINE.json configs 2 and 5 on the GPU (config 1 is the CPU path, tests/test_oracle_cpu.py; configs 3/4
are the bench workload, whose shape is covered by test_full_size_* and the data-parallel tests)."""
import importlib

import numpy as np
import pytest

import np_ref as R

pytestmark = pytest.mark.gpu
aefft = importlib.import_module("autoencoder-fft_amd")


@pytest.fixture(scope="module")
def ctx():
    c = aefft.Context(0)
    yield c
    c.close()


def host(t):
    return t.detach().cpu().numpy()


def _weights(rng, D, maps, Nk, rmax):
    ws, dD = [], D
    for dM in maps:
        q = lambda a: a.astype(np.float32).astype(np.float64)
        ws.append((q(rng.uniform(-rmax, rmax, (dM, dD, Nk, Nk))), q(rng.uniform(-rmax, rmax, dM)),
                   q(rng.uniform(-rmax, rmax, (dD, dM, Nk, Nk))), q(rng.uniform(-rmax, rmax, dD))))
        dD = dM
    return ws


def test_config2_forward_and_step_vs_oracle(ctx):
    """config 2: 256x256, 3 pairs 8/16/32 maps, 5x5, pooling 2 (reference default), FFT mode, B=1."""
    rng = np.random.default_rng(2)
    D, N, maps, Nk, s = 3, 256, [8, 16, 32], 5, 2
    L = len(maps)
    ws = _weights(rng, D, maps, Nk, 3.0)
    x = np.floor(rng.uniform(0, 256, (D, N, N)))
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=1)
    for l, w in enumerate(ws):
        net.set_pair(l, *w)
    net_c = [w[0] for w in ws] + [w[2] for w in ws[::-1]]
    net_b = [w[1] for w in ws] + [w[3] for w in ws[::-1]]
    layers, cfreq, spectra = R.autoenc_fft(x, net_c, net_b, [s] * L + [-s] * L)
    recon = ctx.empty(1, D, N, N)
    net.step_grad(ctx.dev(x[None]), recon)
    for l in (1, 2, 3, 6, 7, 10, 12):
        ref = layers[l]
        assert np.abs(host(net.get_layer(l))[0] - ref).max() < 1e-4 * np.abs(ref).max(), l
    assert np.abs(host(recon)[0] - layers[-1]).max() < 1e-4 * np.abs(layers[-1]).max()
    mse = ctx.empty(L)
    net.step_apply(0.2, 0, 0, 1.0, mse)
    for l in range(L):
        c, b, f, p = ws[l]
        z = lambda a: np.zeros_like(a)
        X = [spectra[2 * l + 1]]; O = [spectra[4 * L - 1 - 2 * l]]
        r = R.batch_train_iter(X, X, O, cfreq[l], cfreq[2 * L - 1 - l], c, f, b, p, (z(c), z(f), z(b), z(p)), 0.02)
        c2, b2, f2, p2 = net.get_pair(l)
        dw = np.abs(r["c"] - c).max()
        for a, k in ((c2, "c"), (f2, "f"), (b2, "b"), (p2, "p")):
            assert np.abs(a - r[k]).max() < 1e-6 + 1e-4 * dw, (l, k)
        assert abs(host(mse)[l] - r["mse"]) < 1e-5 * max(1, r["mse"])
    net.close()


@pytest.mark.parametrize("maxdiff", [0, 1])
def test_tied_weights_fft_mode_small_vs_oracle(ctx, maxdiff):
    """config 5's options (symmetric weights + multiobjective) on a small pair against the oracle's build-defined rule."""
    rng = np.random.default_rng(50 + maxdiff)
    dD, dM, N, Nk, B = 3, 4, 16, 5, 2
    (c, b, f, p), = _weights(rng, dD, [dM], Nk, 1.0)
    f = np.transpose(c, (1, 0, 2, 3)).copy()                      # key 'p' in the reference copies c into f (autoencoder.cpp:343-355)
    xs = np.floor(rng.uniform(0, 256, (B, dD, N, N)))
    net = aefft.Net(ctx, dD, N, N, [dM], Nk, 1, batch=B)
    net.set_pair(0, c, b, f, p)
    net.step_grad(ctx.dev(xs), None)
    mse = ctx.empty(1)
    net.step_apply(0.2, maxdiff, 1, 1.0, mse)
    sp = [R.autoenc_fft(x, [c, f], [b, p], [1, -1]) for x in xs]
    Xs = [s_[2][1] for s_ in sp]; Os = [s_[2][3] for s_ in sp]
    dck, dfk, db, dp = R.batch_grad(Xs, Xs, Os, sp[0][1][0], sp[0][1][1], b, Nk, Nk)
    z = lambda a: np.zeros_like(a)
    extra = R.gradient_diff(c, f, b, p) if maxdiff else (None,) * 4
    rc, rf, rb, rp = R.backprop_sym(c, f, b, p, dck, dfk, db, dp, z(c), z(f), z(b), z(p), 0.02, *extra)[:4]
    c2, b2, f2, p2 = net.get_pair(0)
    dw = np.abs(rc - c).max()
    assert dw > 0
    for a, r in ((c2, rc), (f2, rf), (b2, rb), (p2, rp)):
        assert np.abs(a - r).max() < 1e-6 + 1e-3 * dw
    assert np.array_equal(f2, np.transpose(c2, (1, 0, 2, 3)))
    net.close()


def test_config5_full_size_runs(ctx):
    """config 5 shape: 1024x1024, 5 pairs 8..128 maps, 5x5, pool 2, symmetric + multiobjective, 2 frames per GPU:
    finite MSEs, tied weights hold, and the forward obeys linearity in the frames (biases zero)."""
    rng = np.random.default_rng(5)
    D, N, maps, Nk, s, B = 3, 1024, [8, 16, 32, 64, 128], 5, 2, 2
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    dD = D
    for l, dM in enumerate(maps):
        c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32)
        net.set_pair(l, c, np.zeros(dM, np.float32), np.transpose(c, (1, 0, 2, 3)).copy(), np.zeros(dD, np.float32))
        dD = dM
    x = np.floor(rng.uniform(0, 256, (B, D, N, N))).astype(np.float32)
    r1, r2 = ctx.empty(B, D, N, N), ctx.empty(B, D, N, N)
    net.forward(ctx.dev(x), r1)
    net.forward(ctx.dev(2 * x), r2)
    a1, a2 = host(r1), host(r2)
    assert np.isfinite(a1).all() and np.abs(a2 - 2 * a1).max() < 1e-5 * np.abs(a2).max()
    # distinct biases for the multiobjective step: gradient_diff divides by b[m]-b[m1] (fft.cu:743-746; coinciding
    # biases give inf/nan in the reference too, Appendix B-9)
    for l in range(len(maps)):
        c2, b2, f2, p2 = net.get_pair(l)
        net.set_pair(l, c2, rng.uniform(-1, 1, b2.shape), f2, rng.uniform(-1, 1, p2.shape))
    net.step_grad(ctx.dev(x), r1)
    mse = ctx.empty(len(maps))
    net.step_apply(0.2, 1, 1, 1.0, mse)
    assert np.isfinite(host(mse)).all()
    for l in range(len(maps)):
        c2, b2, f2, p2 = net.get_pair(l)
        assert np.isfinite(c2).all() and np.array_equal(f2, np.transpose(c2, (1, 0, 2, 3)))
    net.close()


LITERAL = ["NOOPFORM", "NOLAZY", "NOCOMPACT", "NOQPATH", "NOFUSEMSE", "NOGROUP", "NOMFMA", "NOGFWD", "NOOVERLAP", "NOFUSECROP"]


@pytest.mark.gpu
def test_optimised_step_equals_literal_sequence_over_three_steps(ctx, flags):
    """Config-2 sized net (256x256, 3 pairs, pool 2), B = 4, three training steps on fresh frames: the default path (matrix
    cores, pooled-grid encoder, support-only decoder, Q-path gradients, collapsed operator, grouped launches) against the
    same library with every one of those switched off, i.e. the literal conv / S / dc,df / C2R / update / R2C / conv, conv /
    MSE sequence on scalar-FMA kernels.  Guards the re-associations at a size the numpy oracle is too slow for."""
    rng = np.random.default_rng(77)
    D, N, maps, Nk, s, B = 3, 256, [8, 16, 32], 5, 2, 4
    ws = _weights(rng, D, maps, Nk, 1.0)
    frames = [ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N)))) for _ in range(3)]
    res = []
    for literal in (True, False):
        flags(*(LITERAL if literal else []))
        net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
        for l, w in enumerate(ws):
            net.set_pair(l, *w)
        recon, mse = ctx.empty(B, D, N, N), ctx.empty(len(maps))
        grads = []
        for x in frames:
            net.step_grad(x, recon)
            grads.append(host(net.grad_buffer()).copy())
            net.step_apply(0.02, 0, 0, 1.0, mse)
        res.append((grads, [net.get_pair(l) for l in range(len(maps))], host(recon).copy(), host(mse).copy()))
        net.close()
    (g_lit, w_lit, r_lit, m_lit), (g_opt, w_opt, r_opt, m_opt) = res
    assert np.abs(g_lit[0] - g_opt[0]).max() < 5e-5 * np.abs(g_lit[0]).max()          # step 1: identical weights on both sides
    for l, (a, b) in enumerate(zip(w_lit, w_opt)):
        for x, y, w0 in zip(a, b, ws[l]):
            dw = np.abs(x - w0).max()
            assert np.abs(x - y).max() < 1e-6 + 2e-3 * dw
    assert np.abs(r_lit - r_opt).max() < 1e-3 * np.abs(r_lit).max()
    assert np.allclose(m_lit, m_opt, rtol=2e-3)
