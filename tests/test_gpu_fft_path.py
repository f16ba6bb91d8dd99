"""GPU parity tests of the FFT-mode path, through the C ABI (include/aefft.h), against the
oracle (oracle/np_ref.py, float64 master) on the same seeded inputs, the committed golden
vectors, and size-independent properties at BASELINE sizes.

Stated float32 tolerances: tensors |d| <= 1e-4*max|ref|; kernels after a step
|d| <= 1e-6 + 1e-4*|dw|... expressed here relative to max|dw|; MSE |d| <= 1e-5*max(1,mse) is
loosened to 1e-4 relative because the MSE itself is a float32 sum of ~1e5 terms."""
import importlib
import os

import numpy as np
import pytest

import np_ref as R

pytestmark = pytest.mark.gpu
aefft = importlib.import_module("autoencoder-fft_amd")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fft_path.npz")
TOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    c = aefft.Context(0)
    yield c
    c.close()


def relerr(got, ref):
    got = np.asarray(got); ref = np.asarray(ref)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------
# transforms
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Nx,Ny,planes", [(8, 8, 3), (16, 8, 2), (8, 32, 5), (32, 32, 7), (64, 64, 3), (128, 64, 2),
                                          (64, 256, 2), (256, 256, 3), (512, 512, 2), (1024, 1024, 1), (2048, 512, 1),
                                          (512, 2048, 1)])
def test_r2c_c2r(ctx, Nx, Ny, planes):
    rng = np.random.default_rng(Nx * 7 + Ny)
    x = np.floor(rng.uniform(0, 256, (planes, Nx, Ny))).astype(np.float32)
    X = ctx.r2c(ctx.dev(x))
    ref = R.fft(x)
    assert relerr(host(X), ref) < 2e-6          # far inside the 1e-4 budget: float32 FFT round-off only
    # inverse of a NOT exactly Hermitian spectrum (gradient spectra are Hermitian only up to rounding)
    Z = ref + 1e-3 * np.abs(ref).max() * (rng.normal(size=ref.shape) + 1j * rng.normal(size=ref.shape))
    y = ctx.c2r(ctx.dev(Z), Ny)
    assert relerr(host(y), R.fft_inv(Z, Nx, Ny)) < 5e-6
    yu = ctx.c2r(ctx.dev(Z), Ny, scale=1.0)     # the unnormalised C2R of fft_backproplib.cu:1219
    assert relerr(host(yu), R.c2r_unnorm(Z, Nx, Ny)) < 5e-6


@pytest.mark.parametrize("N,s", [(32, 2), (64, 4), (256, 2), (512, 2), (128, 8)])
def test_pool_and_fused_forms(ctx, N, s):
    rng = np.random.default_rng(N + s)
    x = np.floor(rng.uniform(0, 256, (3, N, N))).astype(np.float32)
    X = R.fft(x)
    down, nx, ny = R.pool_fft(X, N, N, s)
    Xd, gx, gy = ctx.pool(ctx.dev(X), N, s)
    assert (gx, gy) == (nx, ny) and np.array_equal(host(Xd), down.astype(np.complex64))   # pure index remap: exact
    up, ux, uy = R.pool_fft(down, nx, ny, -s)
    Xu, gx, gy = ctx.pool(ctx.dev(down), ny, -s)
    assert (gx, gy) == (ux, uy) and np.array_equal(host(Xu), up.astype(np.complex64))
    # fused: r2c+crop and zero-pad+c2r
    assert relerr(host(ctx.r2c_pool(ctx.dev(x), s)), down) < 2e-6
    y = ctx.unpool_c2r(ctx.dev(down), ny, -s, 1.0 / (N * N))
    assert relerr(host(y), R.fft_inv(up, N, N)) < 5e-6


def test_pool_rejects_bad_sizes(ctx):
    X = ctx.dev(np.zeros((1, 16, 9), np.complex64))
    with pytest.raises(aefft.AefftError):
        ctx.pool(X, 16, 3)           # int(16 / 3) = 5: an odd grid (the index rules of `resize` are written for even sizes)
    with pytest.raises(aefft.AefftError):
        ctx.pool(X, 16, 4)           # 16/4 < 8
    with pytest.raises(aefft.AefftError):
        ctx.r2c(ctx.dev(np.zeros((1, 13, 12), np.float32)))       # odd size
    with pytest.raises(aefft.AefftError):
        ctx.r2c(ctx.dev(np.zeros((1, 1026, 16), np.float32)))     # not a power of two and beyond the 1024 the chirp-z form serves
    with pytest.raises(aefft.AefftError):
        aefft.Net(ctx, 3, 96, 96, [4], 5, 2, batch=1)             # the resident network: powers of two only


@pytest.mark.parametrize("N,Nk,Nl", [(16, 5, 5), (32, 3, 3), (64, 5, 3), (128, 7, 7)])
def test_kernel_spectrum_and_export(ctx, N, Nk, Nl):
    rng = np.random.default_rng(N + Nk)
    c = rng.uniform(-3, 3, (4, 3, Nk, Nl)).astype(np.float32)
    K = ctx.kernel_spectrum(ctx.dev(c), N, N)
    assert relerr(host(K), R.kernel_spectrum(c, N, N)) < 2e-6
    back = ctx.kernel_export(K, Nk, Nl, N)
    assert np.abs(host(back) - c).max() < 1e-5


# ------------------------------------------------------------------------------------------
# Hadamard contraction, gradient, mse, update
# ------------------------------------------------------------------------------------------
def _pair(rng, dD, dM, N, Nk, B):
    xs = np.floor(rng.uniform(0, 256, (B, dD, N, N)))
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32).astype(np.float64)
    f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32).astype(np.float64)
    b = rng.uniform(-1, 1, dM).astype(np.float32).astype(np.float64)
    p = rng.uniform(-1, 1, dD).astype(np.float32).astype(np.float64)
    return xs, c, f, b, p


@pytest.mark.parametrize("dD,dM,N,B", [(3, 8, 16, 1), (1, 4, 32, 3), (8, 16, 16, 5), (3, 10, 64, 2), (5, 3, 16, 4), (16, 32, 8, 2)])
def test_conv(ctx, dD, dM, N, B):
    rng = np.random.default_rng(dD * 100 + dM)
    xs, c, f, b, p = _pair(rng, dD, dM, N, 5, B)
    X = R.fft(xs); Cs = R.kernel_spectrum(c, N, N)
    ref = np.stack([R.conv_k(X[i], Cs, b, N, N) for i in range(B)])
    O = ctx.conv(ctx.dev(X), ctx.dev(Cs), ctx.dev(b), N)
    assert relerr(host(O), ref) < 5e-6
    # linearity of the contraction (size-independent property)
    O2 = ctx.conv(ctx.dev(2 * X), ctx.dev(Cs), ctx.dev(0 * b), N)
    O0 = ctx.conv(ctx.dev(X), ctx.dev(Cs), ctx.dev(0 * b), N)
    assert relerr(host(O2), 2 * host(O0)) < 1e-6


@pytest.mark.parametrize("dD,dM,N,B", [(3, 4, 16, 1), (2, 5, 32, 1), (3, 8, 16, 4), (8, 16, 16, 3), (1, 3, 64, 2)])
def test_gradient_and_mse(ctx, dD, dM, N, B):
    rng = np.random.default_rng(dD * 31 + dM + N)
    xs, c, f, b, p = _pair(rng, dD, dM, N, 5, B)
    outs = xs + rng.uniform(-20, 20, xs.shape)
    tgt = xs + rng.uniform(-5, 5, xs.shape)                 # expout need not equal in (fft_backproplib.h:11)
    X, T, O = R.fft(xs), R.fft(tgt), R.fft(outs)
    Cs, Fs = R.kernel_spectrum(c, N, N), R.kernel_spectrum(f, N, N)
    per = [R.gradient_k_io(X[i], T[i], O[i], Cs, Fs, b, N, N) for i in range(B)]
    ref = [sum(t) / B for t in zip(*per)]
    got = ctx.gradient(ctx.dev(X), ctx.dev(T), ctx.dev(O), ctx.dev(Cs), ctx.dev(Fs), ctx.dev(b), N)
    for g, r in zip(got, ref):
        # the float32 replay of the reference arithmetic itself sits at 5e-5 here (rounding of the DC bins
        # of O and T before the subtraction), so the stated 1e-4 is the meaningful bound
        assert relerr(host(g), r) < TOL
    mse = ctx.mse(ctx.dev(T), ctx.dev(O), dM, N)
    mref = np.mean([R.mse_fft(T[i], O[i], dM, dD, N, N) for i in range(B)])
    assert abs(host(mse)[0] - mref) < 1e-5 * max(1, mref)


@pytest.mark.parametrize("maxdiff", [0, 1])
@pytest.mark.parametrize("dD,dM,N,Nk", [(3, 4, 16, 5), (2, 3, 32, 3)])
def test_update(ctx, dD, dM, N, Nk, maxdiff):
    rng = np.random.default_rng(5 + dD + maxdiff)
    xs, c, f, b, p = _pair(rng, dD, dM, N, Nk, 1)
    outs = xs + rng.uniform(-20, 20, xs.shape)
    X, O = R.fft(xs[0]), R.fft(outs[0])
    Cs, Fs = R.kernel_spectrum(c, N, N), R.kernel_spectrum(f, N, N)
    dc, df, db, dp = R.gradient_k_io(X, X, O, Cs, Fs, b, N, N)
    z = lambda a: np.zeros_like(a)
    mom = [0.01 * rng.normal(size=a.shape) for a in (c, f, b, p)]
    ref = R.backprop(c, f, b, p, dc, df, db, dp, *mom, N, N, 0.02, maxdiff)
    t = [ctx.dev(a) for a in (c, f, b, p, Cs, Fs, dc, df, db, dp, *mom)]
    ctx.update(*t, N, 0.02, maxdiff)
    names = ["c", "f", "b", "p", "Dc", "Df", "Db", "Dp"]
    got = dict(zip(["c", "f", "b", "p", "C", "F"], t[:6])); got.update(dict(zip(["Dc", "Df", "Db", "Dp"], t[10:])))
    refd = dict(zip(names + ["C", "F"], ref))
    for k in names:
        dw = max(np.abs(refd[k] - dict(c=c, f=f, b=b, p=p, Dc=mom[0], Df=mom[1], Db=mom[2], Dp=mom[3])[k]).max(), 1e-12)
        assert np.abs(host(got[k]) - refd[k]).max() < 1e-6 + 1e-3 * dw, k
    assert relerr(host(got["C"]), refd["C"]) < 5e-6 and relerr(host(got["F"]), refd["F"]) < 5e-6


@pytest.mark.parametrize("Nk,Nl,dD,dM", [(3, 5, 2, 3), (5, 5, 12, 30), (3, 3, 16, 20)])
def test_update_multiobjective_supports_and_chunking(ctx, Nk, Nl, dD, dM):
    """a10 gradient_diff (fft_backproplib.cu:709-753) through aefft_update: a non-square 3x5 support (the generic kernels with the
    stored distance matrix), and channel counts whose dM*dD kernels span several row tiles and partner chunks of the register-tiled
    kernel (360 and 320 kernels: 2 row tiles x 3 chunks), against the oracle's literal loop nest."""
    rng = np.random.default_rng(Nk * 10 + Nl + dM)
    N = 16
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nl)); f = rng.uniform(-1, 1, (dD, dM, Nk, Nl)); b = rng.uniform(-1, 1, dM); p = rng.uniform(-1, 1, dD)
    xs = np.floor(rng.uniform(0, 256, (dD, N, N))); outs = xs + rng.uniform(-20, 20, xs.shape)
    X, O = R.fft(xs), R.fft(outs)
    Cs, Fs = R.kernel_spectrum(c, N, N), R.kernel_spectrum(f, N, N)
    dc, df, db, dp = R.gradient_k_io(X, X, O, Cs, Fs, b, N, N)
    mom = [0.01 * rng.normal(size=a.shape) for a in (c, f, b, p)]
    ref = R.backprop(c, f, b, p, dc, df, db, dp, *mom, N, N, 0.02, 1)
    t = [ctx.dev(a) for a in (c, f, b, p, Cs, Fs, dc, df, db, dp, *mom)]
    ctx.update(*t, N, 0.02, 1)
    start = dict(c=c, f=f, b=b, p=p, Dc=mom[0], Df=mom[1], Db=mom[2], Dp=mom[3])
    got = dict(zip(["c", "f", "b", "p"], t[:4])); got.update(dict(zip(["Dc", "Df", "Db", "Dp"], t[10:])))
    for k, r in zip(["c", "f", "b", "p", "Dc", "Df", "Db", "Dp"], ref):
        dw = max(np.abs(r - start[k]).max(), 1e-12)
        assert np.abs(host(got[k]) - r).max() < 1e-6 + 1e-3 * dw, k


# ------------------------------------------------------------------------------------------
# resident network vs oracle and golden vectors
# ------------------------------------------------------------------------------------------
def _golden_net(ctx, g, tag, D, N, maps, Nk, s, B=1):
    L = len(maps)
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    for l in range(L):
        net.set_pair(l, g[f"{tag}_c{l}"], g[f"{tag}_b{l}"], g[f"{tag}_c{2 * L - 1 - l}"], g[f"{tag}_b{2 * L - 1 - l}"])
    return net


@pytest.mark.parametrize("tag,D,N,maps,Nk,s", [("A", 3, 16, [4], 5, 1), ("B", 3, 32, [4, 6], 3, 2)])
def test_golden_forward_and_bursts(ctx, tag, D, N, maps, Nk, s):
    g = np.load(GOLD)
    L = len(maps)
    x = g[f"{tag}_x"]
    for n_l in range(L):
        for md in (0, 1):
            net = _golden_net(ctx, g, tag, D, N, maps, Nk, s)
            recon = ctx.empty(1, D, N, N)
            net.forward(ctx.dev(x[None]), recon)
            if n_l == 0 and md == 0:
                for l in range(4 * L + 1):
                    ref = g[f"{tag}_layer{l}"]
                    assert relerr(host(net.get_layer(l))[0], ref) < TOL, l
                assert relerr(host(recon)[0], g[f"{tag}_layer{4 * L}"]) < TOL
            mse = net.train_pair(n_l, 3, 0.2, maxdiff=md)
            c, b, f, p = net.get_pair(n_l)
            pre = f"{tag}_burst{n_l}_md{md}_"
            c0 = g[f"{tag}_c{n_l}"]
            dw = np.abs(g[pre + "c"] - c0).max()
            for a, k in ((c, "c"), (f, "f"), (b, "b"), (p, "p")):
                assert np.abs(a - g[pre + k]).max() < 1e-6 + 2e-3 * dw, (k, np.abs(a - g[pre + k]).max(), dw)
            assert np.allclose(mse, g[pre + "mse"], rtol=1e-4), (mse, g[pre + "mse"])
            net.close()


# Every alternative code path of the training step must give the oracle's numbers (development switches, aefft_ctx_set_flags).
STEP_PATHS = ["", "NOOPFORM", "NOOPFORM,GTAPS", "NOOPFORM,NOQPATH", "NOOPFORM,NOCOMPACT", "NOOPFORM,NOLAZY", "NOOPFORM,NOGROUP", "NOOPFORM,NOFUSEMSE",
              "NOOPFORM,NOMFMA", "NOOPFORM,NOOVERLAP", "NOOPFORM,NOFUSECROP", "NOMFMA", "NOGROUP", "NOCHAIN", "NOCOMPACT", "NOLAZY", "NOOVERLAP",
              "NOFUSEUPD", "NOAHEAD", "NOCHAIN,NOFUSEUPD,GTAPS", "CHAINMSE"]


@pytest.mark.parametrize("path", STEP_PATHS)
@pytest.mark.parametrize("B", [1, 3])
def test_step_equals_oracle_batch_iteration(ctx, B, path, flags):
    """aefft_net_step_grad / step_apply == oracle batch_train_iter for every pair (build-defined
    batch mean); B=1 is the reference loop body.  `path` disables one optimisation
    (or forces one that the small test shapes would not choose) so that its fallback is exercised too."""
    flags(*path.split(","))
    _step_vs_oracle(ctx, np.random.default_rng(77 + B), B, 3, 32, 32, [4, 6], 5, 2)


@pytest.mark.parametrize("D,Nx,Ny,maps,Nk,s,B", [
    (3, 32, 32, [4, 6], 3, 2, 2),          # 3x3 kernels: 5x5 offsets in the Q path
    (3, 32, 32, [4], 7, 2, 2),             # 7x7 kernels: no Q path (dc|df spectra route), single pair
    (1, 32, 64, [5, 3], 5, 2, 3),          # non-square planes, gray input, odd map counts
    (3, 64, 64, [4, 6, 5], 5, 2, 2),       # three pairs: decoder chain on the coarsest support
    (3, 32, 32, [4, 6], 5, 1, 2),          # no pooling: nothing to compact, classic S
    (2, 64, 32, [3, 9], 3, 2, 5),          # B not a multiple of 4, Nx > Ny
    (3, 64, 64, [8, 6], 5, 2, 2),          # 8 maps on the outermost pair: the one-round-trip MSE body (opmse_small_body), dD = 3
    (1, 128, 64, [8, 4, 5], 3, 2, 3),      # ... gray input (dD = 1), three pairs, non-square planes
    (2, 128, 128, [8, 16], 5, 2, 1),       # ... dD = 2; pair 0 on 64x64: the inverse transform of S in two row chunks
    (3, 128, 128, [4, 6], 3, 2, 2),        # 3x3 kernels (5x5 offsets) with row chunks: the chunk sum of the 3x3 weight-gradient kernel
    (3, 64, 64, [32, 40, 8], 5, 2, 2),     # a middle pair with dD = 32: rows of G' beyond the 16 requested up front (opmse_gbody), G' tiles of 2 planes
    (3, 64, 64, [128, 128, 128], 5, 2, 2), # 67 steps per chain item: more than the step table holds -- the items run stage by stage (chain_stage_rec)
    (3, 1024, 1024, [4], 5, 2, 2),         # 2056 workgroups in the moments launch: the 4-frame-batch instantiation msgrad_kernel<4> (several rounds of slots)
])
def test_step_shapes_vs_oracle(ctx, D, Nx, Ny, maps, Nk, s, B):
    """the same comparison across kernel supports, plane shapes, depths and pooling settings (edge cases of the tiled paths)"""
    _step_vs_oracle(ctx, np.random.default_rng(5 * Nx + Ny + Nk + B), B, D, Nx, Ny, maps, Nk, s)


def weight_step_tol(gref, del_eff=0.002, grel=5e-5):
    """Per-entry bound on |w_hip - w_oracle| after ONE clipped-momentum step from zero momentum (fft_backproplib.cu:605-652):
    w moves by (1 - alpha) * del * g / max(10, |g|) with (1 - alpha) * del = 0.1 * 0.02 = 0.002.  An entry whose oracle gradient is
    clearly above the knee |g| = 10 moves by exactly 0.002 * sign(g): float32 rounding only (1e-6).  An unclipped entry's step is
    0.0002 * g, so it inherits the gradient tolerance of the comparison above (grel of the largest entry).  Entries within that
    gradient error of the knee get the unclipped bound.  Never more than twice the largest possible step: a net that does not
    apply the update fails."""
    gref = np.asarray(gref, dtype=np.float64)
    gerr = grel * np.abs(gref).max()
    tol = np.where(np.abs(gref) > 10.0 + 2 * gerr, 1e-6, 1e-6 + del_eff / 10.0 * gerr)
    return np.minimum(tol, 2 * del_eff)


def _step_vs_oracle(ctx, rng, B, D, Nx, Ny, maps, Nk, s):
    N = Nx
    L = len(maps)
    xs = np.floor(rng.uniform(0, 256, (B, D, Nx, Ny)))
    ws = []
    dD = D
    for dM in maps:
        _, c, f, b, p = _pair(rng, dD, dM, 8, Nk, 1)
        ws.append((c, b, f, p)); dD = dM
    net = aefft.Net(ctx, D, Nx, Ny, maps, Nk, s, batch=B)
    for l, w in enumerate(ws):
        net.set_pair(l, *w)
    net_c = [w[0] for w in ws] + [w[2] for w in ws[::-1]]
    net_b = [w[1] for w in ws] + [w[3] for w in ws[::-1]]
    sp = [R.autoenc_fft(xs[i], net_c, net_b, [s] * L + [-s] * L) for i in range(B)]
    recon = ctx.empty(B, D, Nx, Ny)
    net.step_grad(ctx.dev(xs), recon)
    for i in range(B):
        assert relerr(host(recon)[i], sp[i][0][-1]) < TOL
    gbuf = host(net.grad_buffer()).copy()
    mse = ctx.empty(L)
    net.step_apply(0.2, 0, 0, 1.0, mse)
    off = 0
    for l in range(L):
        c, b, f, p = ws[l]
        dM, dDl = c.shape[:2]
        Xs = [sp[i][2][2 * l + 1] for i in range(B)]
        Os = [sp[i][2][4 * L - 1 - 2 * l] for i in range(B)]
        cf = sp[0][1]
        z = lambda a: np.zeros_like(a)
        r = R.batch_train_iter(Xs, Xs, Os, cf[l], cf[2 * L - 1 - l], c, f, b, p, (z(c), z(f), z(b), z(p)), 0.02)
        nk = c.size
        for seg, ref in zip((gbuf[off:off + nk], gbuf[off + nk:off + 2 * nk], gbuf[off + 2 * nk:off + 2 * nk + dM],
                             gbuf[off + 2 * nk + dM:off + 2 * nk + dM + dDl]), r["grads"]):
            assert relerr(seg, ref.ravel()) < 5e-5
        off += 2 * nk + dM + dDl
        c2, b2, f2, p2 = net.get_pair(l)
        for (a, k), gref in zip(((c2, "c"), (f2, "f"), (b2, "b"), (p2, "p")), r["grads"]):
            assert (np.abs(a - r[k]) < weight_step_tol(gref)).all(), (k, np.abs(a - r[k]).max())
        assert np.abs(c2 - c).max() > 1e-4 and np.abs(f2 - f).max() > 1e-4, "the update was not applied"
        assert abs(host(mse)[l] - r["mse"]) < 1e-5 * max(1, r["mse"])
    net.close()


@pytest.mark.parametrize("path", ["", "NOAHEAD", "NOCHAIN", "NOFUSEUPD", "NOOPFORM", "NOOPFORM,NOGFWD", "NOOPFORM,NOCOMPACT", "NOOPFORM,NOMFMA", "NOCOMPACT"])
def test_second_step_equals_fresh_net_with_updated_weights(ctx, path, flags):
    """State carried from one training step to the next (the collapsed operator G of the innermost pair, cached spectra,
    stale-layer flags) must be invisible: step 2 on a live net == step 1 of a fresh net that starts from the live net's
    weights and momentum-free... momentum persists across steps, so the comparison is on what does not depend on it:
    the reconstruction, every layer and the packed gradients of step 2."""
    flags(*path.split(","))
    rng = np.random.default_rng(123)
    D, N, maps, Nk, s, B = 3, 64, [4, 6, 5], 5, 2, 3
    L = len(maps)
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    dD = D
    for l, dM in enumerate(maps):
        _, cw, fw, bw, pw = _pair(rng, dD, dM, 8, Nk, 1)
        net.set_pair(l, cw, bw, fw, pw); dD = dM
    x1 = ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N))))
    x2 = ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N))))
    r_live = ctx.empty(B, D, N, N)
    net.step_grad(x1, None); net.step_apply(0.2)
    weights = [net.get_pair(l) for l in range(L)]
    net.step_grad(x2, r_live)
    g_live = host(net.grad_buffer()).copy()
    layers_live = [host(net.get_layer(l)).copy() for l in range(1, 4 * L + 1)]
    fresh = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    for l, (cw, bw, fw, pw) in enumerate(weights):
        fresh.set_pair(l, cw, bw, fw, pw)
    r_fresh = ctx.empty(B, D, N, N)
    fresh.step_grad(x2, r_fresh)
    g_fresh = host(fresh.grad_buffer())
    assert relerr(host(r_live), host(r_fresh)) < 2e-5
    # the last L floats of the packed buffer carry the PREVIOUS step's post-update MSEs (zero on a fresh net): not compared
    assert relerr(g_live[:-L], g_fresh[:-L]) < 5e-5
    for a, b in zip(layers_live, [host(fresh.get_layer(l)) for l in range(1, 4 * L + 1)]):
        assert relerr(a, b) < 5e-5
    net.close(); fresh.close()


def test_input_prefetch_gives_identical_training(ctx):
    """aefft_net_set_input_ready: the input R2C on a side stream with double-buffered input spectra only reorders work;
    five steps with fresh frames each must leave exactly the weights of the stream-ordered run."""
    rng = np.random.default_rng(321)
    D, N, maps, Nk, s, B = 3, 64, [4, 6], 5, 2, 4
    ws = []
    dD = D
    for dM in maps:
        _, cw, fw, bw, pw = _pair(rng, dD, dM, 8, Nk, 1)
        ws.append((cw, bw, fw, pw)); dD = dM
    frames = [ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N)))) for _ in range(5)]
    out = []
    for ready in (False, True):
        net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
        for l, w in enumerate(ws):
            net.set_pair(l, *w)
        net.set_input_ready(ready)
        recon, mse = ctx.empty(B, D, N, N), ctx.empty(len(maps))
        for x in frames:
            net.step_grad(x, recon); net.step_apply(0.2, 0, 0, 1.0, mse)
        ctx.sync()
        out.append(([net.get_pair(l) for l in range(len(maps))], host(recon).copy(), host(mse).copy()))
        net.close()
    for wa, wb in zip(out[0][0], out[1][0]):
        for a, b in zip(wa, wb):
            assert np.array_equal(a, b)
    assert np.array_equal(out[0][1], out[1][1])
    assert np.allclose(out[0][2], out[1][2], rtol=1e-5)


def test_spectra_store_load_roundtrip(ctx):
    """net_cfreq semantics: store_cfreq / load_cfreq (fft_backproplib.cu:1117-1141)."""
    rng = np.random.default_rng(3)
    _, c, f, b, p = _pair(rng, 3, 4, 16, 5, 1)
    net = aefft.Net(ctx, 3, 16, 16, [4], 5, 1, batch=1)
    net.set_pair(0, c, b, f, p)
    Cs, Fs = net.store_spectra(0)
    assert relerr(Cs, R.kernel_spectrum(c, 16, 16)) < 2e-6 and relerr(Fs, R.kernel_spectrum(f, 16, 16)) < 2e-6
    net2 = aefft.Net(ctx, 3, 16, 16, [4], 5, 1, batch=1)
    net2.load_spectra(0, Cs, b, Fs, p)
    c2, b2, f2, p2 = net2.get_pair(0)
    assert np.abs(c2 - c).max() < 1e-5 and np.abs(f2 - f).max() < 1e-5 and np.array_equal(b2, b.astype(np.float32))
    x = ctx.dev(np.floor(rng.uniform(0, 256, (1, 3, 16, 16))))
    r1, r2 = ctx.empty(1, 3, 16, 16), ctx.empty(1, 3, 16, 16)
    net.forward(x, r1); net2.forward(x, r2)
    assert np.array_equal(host(r1), host(r2))
    net.close(); net2.close()


def test_train_requires_forward(ctx):
    net = aefft.Net(ctx, 3, 16, 16, [4], 5, 1, batch=1)
    with pytest.raises(aefft.AefftError):
        net.train_pair(0, 1, 0.2)
    with pytest.raises(aefft.AefftError):
        net.step_apply(0.2)
    net.close()


# ------------------------------------------------------------------------------------------
# BASELINE-size properties (no oracle run at these sizes)
# ------------------------------------------------------------------------------------------
def test_full_size_roundtrip_parseval_and_delta_identity(ctx):
    """512x512, 4 pairs (cfg3 shape, B=2): r2c->c2r identity, Parseval, and delta kernels make
    the s=1 autoencoder the identity divided by prod(dM*dD) (conv_k divides by the output count)."""
    import torch
    rng = np.random.default_rng(9)
    N, B, D = 512, 2, 3
    x = np.floor(rng.uniform(0, 256, (B, D, N, N))).astype(np.float32)
    xd = ctx.dev(x)
    X = ctx.r2c(xd)
    assert relerr(host(ctx.c2r(X, N)), x) < 2e-6
    Xh = host(X).astype(np.complex128)
    w = np.full(N // 2 + 1, 2.0); w[0] = w[-1] = 1.0
    assert abs((np.abs(Xh) ** 2 * w).sum() / (N * N) / (x.astype(np.float64) ** 2).sum() - 1) < 1e-6
    maps = [8, 16, 32, 64]
    net = aefft.Net(ctx, D, N, N, maps, 5, 2, batch=B)
    dD = D; fac = 1.0
    for l, dM in enumerate(maps):
        c = np.zeros((dM, dD, 5, 5), np.float32); f = np.zeros((dD, dM, 5, 5), np.float32)
        for d in range(dD):
            c[d, d, 2, 2] = 1; f[d, d, 2, 2] = 1           # channel d passes through map d
        net.set_pair(l, c, np.zeros(dM, np.float32), f, np.zeros(dD, np.float32))
        fac *= dM * dD; dD = dM
    recon = ctx.empty(B, D, N, N)
    net.forward(xd, recon)
    # band-limit the reference the same way the 4 poolings do: keep the lowest 32x32 frequencies
    Xl = R.fft(x[0].astype(np.float64))
    low, nx, ny = Xl, N, N
    for _ in maps:
        low, nx, ny = R.pool_fft(low, nx, ny, 2)
    for _ in maps:
        low, nx, ny = R.pool_fft(low, nx, ny, -2)
    ref = R.fft_inv(low, N, N) / fac
    assert relerr(host(recon)[0], ref) < TOL
    net.close()
