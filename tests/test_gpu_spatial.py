"""GPU parity tests of the spatial-mode path (Conv_gpu / backprop_gpu / backprop_gpu_cc semantics)
through the C ABI against oracle/np_spatial.py (cross-pinned to the compiled CPU reference)."""
import importlib

import numpy as np
import pytest

import cpu
import np_spatial as S

pytestmark = pytest.mark.gpu
aefft = importlib.import_module("autoencoder-fft_amd")


@pytest.fixture(scope="module")
def ctx():
    c = aefft.Context(0)
    yield c
    c.close()


def host(t):
    return t.detach().cpu().numpy()


def _case(rng, dD, dM, N, Nk, B):
    x = np.floor(rng.uniform(0, 256, (B, dD, N, N))).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    b = rng.uniform(-1, 1, dM).astype(np.float32); p = rng.uniform(-1, 1, dD).astype(np.float32)
    return x, c, b, f, p


@pytest.mark.parametrize("dD,dM,N,Nk,B", [(1, 4, 128, 3, 1), (3, 10, 32, 5, 2), (2, 3, 17, 7, 1), (8, 16, 16, 5, 3)])
def test_conv_gpu_semantics(ctx, dD, dM, N, Nk, B):
    x, c, b, f, p = _case(np.random.default_rng(dD + dM + N), dD, dM, N, Nk, B)
    got = host(ctx.conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b)))
    for i in range(B):
        ref = S.conv(x[i], c, b)
        assert np.abs(got[i] - ref).max() < 1e-5 * max(1, np.abs(ref).max())


@pytest.mark.parametrize("dD,dM,N,Nk,B,sem", [(50, 3, 64, 3, 2, "gpu"), (6, 1, 128, 5, 1, "gpu"), (9, 2, 96, 3, 1, "cpu"), (5, 4, 72, 5, 2, "gpu"),
                                              (20, 3, 80, 7, 1, "cpu")])
def test_conv_few_maps_register_blocked_kernel(ctx, flags, dD, dM, N, Nk, B, sem):
    """dconv4_kernel (<= 4 output maps, planes >= 64 wide): against the oracle and against the tile kernel (NOFAST), channel
    counts that are not a multiple of the 4 it prefetches, widths that are not a multiple of its 64-column tile, both semantics."""
    x, c, b, f, p = _case(np.random.default_rng(dD * 3 + dM + N), dD, dM, N, Nk, B)
    flags()
    got = host(ctx.conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b), semantics=sem))
    flags("NOFAST")
    old = host(ctx.conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b), semantics=sem))
    flags()
    scale = max(1.0, np.abs(old).max())
    assert np.abs(got - old).max() < 2e-5 * scale
    if sem == "gpu":
        for i in range(B):
            ref = S.conv(x[i], c, b)
            assert np.abs(got[i] - ref).max() < 1e-5 * max(1, np.abs(ref).max())


def test_conv_cpu_semantics_matches_compiled_reference(ctx):
    """cpu_semantics=1 reproduces netlib.cpp Conv; checked against the compiled reference when present."""
    L = cpu.reference() or cpu.port()
    x, c, b, f, p = _case(np.random.default_rng(4), 3, 4, 20, 5, 1)
    got = host(ctx.conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b), semantics="cpu"))[0]
    ref = L.conv(x[0], c, b)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("tied", [False, True])
@pytest.mark.parametrize("dD,dM,N,Nk,B", [(1, 4, 32, 3, 1), (3, 5, 16, 5, 1), (2, 3, 16, 3, 3)])
def test_backprop_gpu(ctx, dD, dM, N, Nk, B, tied):
    rng = np.random.default_rng(dD * 7 + dM + int(tied))
    x, c, b, f, p = _case(rng, dD, dM, N, Nk, B)
    hin = np.stack([S.conv(x[i], c, b) for i in range(B)]).astype(np.float32)
    out = np.stack([S.conv(hin[i], f, p) for i in range(B)]).astype(np.float32)
    mom = [0.01 * rng.normal(size=a.shape).astype(np.float32) for a in (c, b, f, p)]       # dc, db, df, dp
    ref = S.backprop_gpu(list(x), list(out), list(hin), c, b, f, p, mom[0], mom[1], mom[2], mom[3], 0.2, 0.9,
                         tied=tied, B_mean=True)
    t = [ctx.dev(a) for a in (x, out, hin, c, b, f, p)]
    tm = [ctx.dev(a) for a in mom]
    tg = [ctx.dev(np.zeros_like(a)) for a in (c, b, f, p)]
    ctx.backprop_spatial(*t, tm, tg, 0.2, 0.9, tied=tied)
    names = ["c", "b", "f", "p", "dc", "db", "df", "dp", "ddc", "ddb", "ddf", "ddp"]
    got = dict(zip(names, [host(a) for a in (t[3], t[4], t[5], t[6], *tm, *tg)]))
    refd = dict(zip(names, ref))
    for k in names:
        if refd[k] is None or (tied and k == "df"):
            continue
        scale = max(np.abs(refd[k]).max(), 1e-6) if k.startswith("d") else max(np.abs(refd["dc"]).max(), 1e-6)
        assert np.abs(got[k] - refd[k]).max() < 1e-6 + 1e-3 * scale, k
    if tied:
        assert np.array_equal(got["f"], np.transpose(got["c"], (1, 0, 2, 3)))    # backproplib.cu:622


def test_gradient_cpu_semantics_vs_compiled_reference(ctx):
    """lo=1 / CPU geometry: the device gradient equals what the compiled CPU backprop applies."""
    L = cpu.reference() or cpu.port()
    rng = np.random.default_rng(21)
    dD, dM, N, Nk = 2, 3, 12, 5
    x = rng.uniform(0, 16, (1, dD, N, N)).astype(np.float32); out = (x + rng.uniform(-2, 2, x.shape)).astype(np.float32)
    hin = rng.uniform(-4, 4, (1, dM, N, N)).astype(np.float32); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    dele = 1e-12
    z = np.zeros((dM, dD, Nk, Nk), np.float32)
    c2, b2, f2, p2 = L.backprop(x[0], out[0], hin[0], z, np.zeros(dM, np.float32), f, np.zeros(dD, np.float32), dele)
    t = [ctx.dev(a) for a in (x, out, hin, z, np.zeros(dM), f, np.zeros(dD))]
    tm = [ctx.dev(np.zeros_like(a)) for a in (z, np.zeros(dM), f, np.zeros(dD))]
    tg = [ctx.dev(np.zeros_like(a)) for a in (z, np.zeros(dM), f, np.zeros(dD))]
    ctx.backprop_spatial(*t, tm, tg, 0.0, 0.0, semantics="cpu")
    ref = -c2.astype(np.float64) * 10 / dele
    assert np.abs(host(tg[0]) - ref).max() < 1e-4 * np.abs(ref).max()
    refb = -b2.astype(np.float64) * 10 / dele
    assert np.abs(host(tg[1]) - refb).max() < 1e-4 * np.abs(refb).max()


@pytest.mark.parametrize("N,scale", [(32, 2), (33, 2), (20, 4), (16, 1), (16, -2), (9, -3)])
def test_pool_on_device_is_bit_identical_to_the_reference_pool(ctx, N, scale):
    """aefft_pool_spatial == netlib.cpp Pool (compiled reference when present, else its C port): integer accumulator
    (truncation, clamp at 0, also at scale 1) going down, nearest neighbour going up; ragged sizes included."""
    L = cpu.reference() or cpu.port()
    rng = np.random.default_rng(100 + N + scale)
    D = 3
    x = (rng.uniform(-40, 260, (D, N, N))).astype(np.float32)
    if scale > 0:
        shape = (D, (N + scale - 1) // scale, (N + scale - 1) // scale)
    else:
        shape = (D, N * -scale, N * -scale)
    ref = L.pool(x, shape, scale)
    got = host(ctx.pool_spatial(ctx.dev(x), shape[1:], scale))
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("dD,dM,N,Nk,B", [(3, 50, 64, 3, 2), (3, 10, 48, 5, 1), (10, 50, 32, 3, 1), (16, 32, 40, 5, 2), (4, 70, 24, 7, 1)])
def test_conv_matrix_core_kernel_vs_oracle_and_vs_valu_kernel(ctx, dD, dM, N, Nk, B):
    """a13 as an implicit GEMM on the matrix cores (mconv_kernel: M = maps, N = pixels, K = dD*Nk*Nl): against the oracle, and
    against the VALU tile kernel (NOMFMA) -- two independent implementations of the same sums; map counts that are not multiples
    of 16, K not a multiple of 4, several channel chunks, ragged tiles."""
    x, c, b, f, p = _case(np.random.default_rng(dD * 3 + dM + N), dD, dM, N, Nk, B)
    got = host(ctx.conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b)))
    ctx.set_flags("NOMFMA")
    try:
        valu = host(ctx.conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b)))
    finally:
        ctx.set_flags()
    for i in range(B):
        ref = S.conv(x[i], c, b)
        assert np.abs(got[i] - ref).max() < 1e-5 * max(1, np.abs(ref).max())
    assert np.abs(got - valu).max() < 1e-5 * np.abs(valu).max()


@pytest.mark.parametrize("dD,dM,N,Nk,s,sem", [(3, 10, 64, 3, 2, "gpu"), (1, 4, 64, 5, 4, "gpu"), (3, 20, 32, 3, 1, "cpu"), (2, 9, 48, 7, 2, "cpu")])
def test_fused_pool_conv_equals_pool_then_conv(ctx, dD, dM, N, Nk, s, sem):
    """Pool + Conv in one launch == the reference's Pool (compiled, integer accumulator) followed by the conv, and
    the pooled layer it publishes is bit-identical to Pool's."""
    L = cpu.reference() or cpu.port()
    rng = np.random.default_rng(N + Nk + s)
    B = 2
    x = rng.uniform(-40, 260, (B, dD, N, N)).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32); b = rng.uniform(-1, 1, dM).astype(np.float32)
    pooled, out = ctx.pool_conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b), s, semantics=sem)
    n = N // s
    for i in range(B):
        pref = L.pool(x[i], (dD, n, n), s)
        assert np.array_equal(host(pooled)[i], pref)
        ref = S.conv(pref, c, b, cpu_semantics=(sem == "cpu"))
        assert np.abs(host(out)[i] - ref).max() < 1e-5 * max(1, np.abs(ref).max())
    # and without publishing the pooled layer
    _, out2 = ctx.pool_conv_spatial(ctx.dev(x), ctx.dev(c), ctx.dev(b), s, semantics=sem, want_pooled=False)
    assert np.array_equal(host(out2), host(out))


@pytest.mark.parametrize("dD,dM,N,Nk,B,tied", [(1, 4, 128, 3, 1, False), (3, 8, 256, 3, 2, False), (3, 6, 128, 5, 1, True)])
def test_backprop_gpu_many_bands(ctx, dD, dM, N, Nk, B, tied):
    """a14/a15 at config-1 size and beyond: 8-16 row bands per frame, several column tiles, several frames -- the partial-sum
    buffers of the correlation kernels and their fixed-order reduction (wsum_kernel) against the oracle."""
    rng = np.random.default_rng(N + Nk + dM)
    x, c, b, f, p = _case(rng, dD, dM, N, Nk, B)
    hin = np.stack([S.conv(x[i], c, b) for i in range(B)]).astype(np.float32)
    out = np.stack([S.conv(hin[i], f, p) for i in range(B)]).astype(np.float32)
    mom = [np.zeros_like(a) for a in (c, b, f, p)]
    ref = S.backprop_gpu(list(x), list(out), list(hin), c, b, f, p, mom[0], mom[1], mom[2], mom[3], 0.2, 0.9, tied=tied, B_mean=True)
    t = [ctx.dev(a) for a in (x, out, hin, c, b, f, p)]
    tm = [ctx.dev(a) for a in mom]
    tg = [ctx.dev(np.zeros_like(a)) for a in (c, b, f, p)]
    ctx.backprop_spatial(*t, tm, tg, 0.2, 0.9, tied=tied)
    names = ["ddc", "ddb", "ddf", "ddp"]
    for k, g, r in zip(names, tg, ref[8:]):
        if r is None:
            continue
        assert np.abs(host(g) - r).max() < 2e-5 * max(np.abs(r).max(), 1e-30), k


@pytest.mark.parametrize("dD,dM,N,Nk", [(2, 2, 8, 3), (2, 3, 12, 5), (1, 2, 10, 3)])
def test_b11_compat_switch_reproduces_the_cuda_source(ctx, dD, dM, N, Nk):
    """semantics="cuda_compat" gives the decoder-kernel / encoder-bias gradients of the CUDA source as
    written (hidden layer read at (i-ik)*Nx + (j-ik), stale per-pixel buffer, `dDdB2 =`), i.e. the literal restatement
    oracle/np_spatial_literal.py with compat=True; the other two gradients are untouched."""
    import np_spatial_literal as SL
    rng = np.random.default_rng(dD + dM + N + Nk)
    x = rng.uniform(0, 16, (1, dD, N, N)).astype(np.float32); out = (x + rng.uniform(-2, 2, x.shape)).astype(np.float32)
    hin = rng.uniform(-4, 4, (1, dM, N, N)).astype(np.float32); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    z = np.zeros((dM, dD, Nk, Nk), np.float32)
    res = {}
    for sem in ("gpu", "cuda_compat"):
        t = [ctx.dev(a) for a in (x, out, hin, z, np.zeros(dM), f, np.zeros(dD))]
        tm = [ctx.dev(np.zeros_like(a)) for a in (z, np.zeros(dM), f, np.zeros(dD))]
        tg = [ctx.dev(np.zeros_like(a)) for a in (z, np.zeros(dM), f, np.zeros(dD))]
        ctx.backprop_spatial(*t, tm, tg, 0.0, 0.0, semantics=sem)
        res[sem] = [host(g) for g in tg]                          # ddc, ddb, ddf, ddp
    for sem, compat in (("gpu", False), ("cuda_compat", True)):
        gc, gf, gb, gp = SL.gradients_literal(x[0], out[0], hin[0], f, compat=compat)
        for got, ref, k in zip(res[sem], (gc, gb, gf, gp), ("gc", "gb", "gf", "gp")):
            assert np.abs(got - ref).max() < 2e-5 * max(np.abs(ref).max(), 1e-30), (sem, k)
    assert not np.allclose(res["gpu"][2], res["cuda_compat"][2])


@pytest.mark.parametrize("dD,dM,N,B,tied,sem", [(3, 8, 24, 2, False, "gpu"), (1, 9, 40, 1, False, "gpu"), (3, 50, 16, 3, True, "gpu"),
                                                (3, 8, 264, 1, False, "gpu"), (3, 12, 36, 2, False, "cpu"), (1, 8, 20, 1, True, "cpu")])
def test_backprop_kernel_gradient_through_the_error_input_correlation(ctx, flags, dD, dM, N, B, tied, sem):
    """a14/a15, 3x3 supports with <= 3 input channels: dC and dB through the region sums of the error-input correlation (rcorr_kernel:
    no back-convolved error tensor) == the route through the back-convolved error (switch NORCORR) == the oracle -- border rows and
    columns included (rows 0, 1, N-1 are their own regions), several row bands, two column blocks (N = 264), both boundary semantics,
    and with a hidden layer that is NOT conv(in): the hidden layer enters dF only, as the caller passed it."""
    rng = np.random.default_rng(N + dM + dD)
    x, c, b, f, p = _case(rng, dD, dM, N, 3, B)
    hin = rng.uniform(-50, 50, (B, dM, N, N)).astype(np.float32)
    out = (x + rng.uniform(-20, 20, x.shape)).astype(np.float32)
    mom = [np.zeros_like(a) for a in (c, b, f, p)]
    res = []
    for fl in ((), ("NORCORR",)):
        flags(*fl)
        t = [ctx.dev(a) for a in (x, out, hin, c, b, f, p)]
        tm = [ctx.dev(a) for a in mom]
        tg = [ctx.dev(np.zeros_like(a)) for a in (c, b, f, p)]
        ctx.backprop_spatial(*t, tm, tg, 0.2, 0.9, tied=tied, semantics=sem)
        res.append([host(g).copy() for g in tg])
    for k, (g0, g1) in zip(("ddc", "ddb", "ddf", "ddp"), zip(*res)):
        assert np.abs(g0 - g1).max() <= 3e-5 * max(np.abs(g1).max(), 1e-30), k
    if sem == "gpu":
        ref = S.backprop_gpu(list(x), list(out), list(hin), c, b, f, p, mom[0], mom[1], mom[2], mom[3], 0.2, 0.9, tied=tied, B_mean=True)
        for k, g, r in zip(("ddc", "ddb", "ddf", "ddp"), res[0], ref[8:]):
            if r is not None:
                assert np.abs(g - r).max() < 3e-5 * max(np.abs(r).max(), 1e-30), k


@pytest.mark.parametrize("dD,dM,N,B,tied,sem", [(3, 50, 40, 2, False, "gpu"), (3, 16, 264, 1, False, "gpu"), (1, 8, 64, 3, True, "gpu"), (3, 12, 48, 2, True, "gpu"),
                                                (3, 10, 40, 2, False, "cpu"), (3, 9, 32, 1, True, "cpu"), (3, 6, 32, 2, False, "gpu"), (4, 10, 32, 1, False, "gpu")])
def test_fused_spatial_step_gradients_from_the_region_sums(ctx, flags, dD, dM, N, B, tied, sem):
    """aefft_step_spatial = Conv_gpu, Conv_gpu, backprop_gpu[_cc] in one call.  The hidden layer is then the call's own convolution of
    the input, and dF, dP are contracted from the same error-input region sums as dC, dB (the hidden layer is not read by the gradient):
    layers, gradients, updated weights and momentum == the three separate calls (which read the hidden layer) == the oracle; both
    boundary semantics, tied weights, border regions, two column blocks (N = 264), and shapes the region route declines (dM = 6 < 8
    maps, 4 input channels), which run the plain sequence inside the same call."""
    rng = np.random.default_rng(7 * N + dM + dD)
    x, c, b, f, p = _case(rng, dD, dM, N, 3, B)
    mom = [0.01 * rng.normal(size=a.shape).astype(np.float32) for a in (c, b, f, p)]
    flags()
    # the separate calls
    tw = [ctx.dev(a) for a in (c, b, f, p)]
    tm = [ctx.dev(a) for a in mom]
    tg = [ctx.dev(np.zeros_like(a)) for a in (c, b, f, p)]
    xd = ctx.dev(x)
    h0 = ctx.conv_spatial(xd, tw[0], tw[1], semantics=sem)
    o0 = ctx.conv_spatial(h0, tw[2], tw[3], semantics=sem)
    ctx.backprop_spatial(xd, o0, h0, *tw, tm, tg, 0.2, 0.9, tied=tied, semantics=sem)
    sep = [host(t).copy() for t in tw + tm + tg]
    # the fused call
    tw = [ctx.dev(a) for a in (c, b, f, p)]
    tm = [ctx.dev(a) for a in mom]
    tg = [ctx.dev(np.zeros_like(a)) for a in (c, b, f, p)]
    h1, o1 = ctx.step_spatial(xd, *tw, tm, tg, 0.2, 0.9, tied=tied, semantics=sem)
    fus = [host(t).copy() for t in tw + tm + tg]
    assert np.array_equal(host(h1), host(h0)) and np.array_equal(host(o1), host(o0))
    names = ["c", "b", "f", "p", "dc", "db", "df", "dp", "ddc", "ddb", "ddf", "ddp"]
    for k, a, r in zip(names, fus, sep):
        if tied and k in ("df", "ddf"):
            continue
        ref_scale = max(np.abs(sep[8]).max(), 1e-30) if k in ("ddc", "ddf") else max(np.abs(r).max(), 1e-30)
        if k.startswith("dd"):
            assert np.abs(a - r).max() < 5e-5 * ref_scale, (k, np.abs(a - r).max(), ref_scale)
        else:
            # one clipped-momentum step from the same gradients up to 5e-5: 0.02 * dg / 10
            assert np.abs(a - r).max() < 1e-6 + 0.02 / 10 * 5e-5 * max(np.abs(sep[8]).max(), np.abs(sep[10]).max(), np.abs(sep[9]).max(), np.abs(sep[11]).max()), k
    if sem == "gpu":
        hin = [S.conv(x[i], c, b) for i in range(B)]
        out = [S.conv(hin[i], f, p) for i in range(B)]
        ref = S.backprop_gpu(list(x), out, hin, c, b, f, p, mom[0], mom[1], mom[2], mom[3], 0.2, 0.9, tied=tied, B_mean=True)
        for k, g, r in zip(("ddc", "ddb", "ddf", "ddp"), fus[8:], ref[8:]):
            if r is not None:
                assert np.abs(g - r).max() < 5e-5 * max(np.abs(r).max(), 1e-30), k
