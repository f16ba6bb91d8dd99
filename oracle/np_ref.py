"""TEST INFRASTRUCTURE ONLY -- not product code.

numpy restatement of the reference's FFT-mode ("momentum space") training path,
`/root/reference/source/fft_backproplib.cu`.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module; the product path
(`autoencoder-fft_amd/csrc`) never does.

Parity status: the reference ships no tests, fixtures or known-answer vectors, and its
FFT path is CUDA + cuFFT (CUDA toolkit 11.7, `install-dependencies.sh:17`), which cannot
be built or run in this image.  cuFFT R2C/C2R is restated as the published unnormalised
DFT pair (== numpy.fft.rfft2 / irfft2 * Nx*Ny).  The restatement is therefore
**parity unpinned** by reference-held vectors; it is cross-pinned instead against the
reference's own CPU code compiled from its sources (`oracle/_ref`, see
`tests/test_oracle_crosspin.py`): FFT-mode forward == compiled `Conv` on interior pixels
for 3x3 kernels, FFT-mode weight gradient == compiled `backprop` update direction on
zero-margin inputs, plus finite-difference checks of the gradient formulas.

Every function cites the reference lines it follows (`fft.cu` = fft_backproplib.cu).
Layouts are the reference's: activations `[ch][Nx][Ny]`, spectra `[ch][Nx][Nyr]`
(Nyr = Ny/2+1, complex), encoder kernels `c[dM][dD][Nk][Nl]`, decoder kernels
`f[dD][dM][Nk][Nl]`.

`dtype` selects the replay precision: np.float64 is the master, np.float32 replays the
reference's float arithmetic (scipy.fft keeps float32).
"""
import numpy as np
import scipy.fft as sfft


def _ctype(dtype):
    return np.complex64 if np.dtype(dtype) == np.float32 else np.complex128


# ----------------------------------------------------------------------------------------
# cuFFT call sites (fft.cu:764-801 `fft`, 806-864 `fft_inv`, 869-916 `kfft`, 921-970 `kfft_inv`)
# ----------------------------------------------------------------------------------------
def fft(x, dtype=np.float64):
    """fft.cu:764-801: batched unnormalised 2-D R2C over the last two axes."""
    x = np.asarray(x, dtype=dtype)
    return sfft.rfft2(x, axes=(-2, -1)).astype(_ctype(dtype), copy=False)


def c2r_unnorm(X, Nx, Ny, dtype=np.float64):
    """cufftExecC2R (fft.cu:829,946,1219-1220): unnormalised inverse.  Imaginary parts of the
    self-conjugate bins are ignored (pocketfft semantics; cuFFT's are unspecified for
    non-Hermitian input -- compare with tolerance)."""
    X = np.asarray(X, dtype=_ctype(dtype))
    y = sfft.irfft2(X, s=(Nx, Ny), axes=(-2, -1))
    return (y * dtype(Nx * Ny)).astype(dtype, copy=False)


def fft_inv(X, Nx, Ny, dtype=np.float64):
    """fft.cu:806-864: C2R then host multiply by norm=1/(Nx*Ny) (`:831,841`)."""
    norm = dtype(1.0) / dtype(Nx * Ny)
    return (c2r_unnorm(X, Nx, Ny, dtype) * norm).astype(dtype, copy=False)


# ----------------------------------------------------------------------------------------
# spectral pooling (fft.cu:87-157 `resize`, 975-1002 `pool_fft`)
# ----------------------------------------------------------------------------------------
def pooled_size(Nx, Ny, scale):
    """fft.cu:980-984: l = scale or 1/|scale| as float; Nxs = int(Nx/l)."""
    l = np.float32(scale)
    if scale < 0:
        l = np.float32(-1.0) / np.float32(scale)
    return int(np.float32(Nx) / l), int(np.float32(Ny) / l)


def resize(freq, Nx, Ny, Nxs, Nys):
    """fft.cu:87-157, literal index remap (even sizes).  freq: [ch][Nx][Nyr] -> [ch][Nxs][Nyrs];
    destination is zero-initialised (`fft.cu:990`); no amplitude rescale (`:154-155` commented)."""
    freq = np.asarray(freq)
    Nyr, Nyrs = Ny // 2 + 1, Nys // 2 + 1
    out = np.zeros(freq.shape[:-2] + (Nxs, Nyrs), dtype=freq.dtype)
    for i in range(Nxs):
        if Nxs <= Nx:
            if i < Nxs // 2:
                si = i
            elif i == Nxs // 2:
                si = Nx // 2
            else:
                si = i + Nx - Nxs
            # j < Nyrs-1 -> same column; j == Nyrs-1 -> source Nyquist column (fft.cu:100-113)
            out[..., i, : Nyrs - 1] = freq[..., si, : Nyrs - 1]
            out[..., i, Nyrs - 1] = freq[..., si, Nyr - 1]
        else:
            if i < Nx // 2:
                si = i
            elif i > Nxs - Nx // 2:
                si = i - Nxs + Nx
            elif i == Nxs // 2:
                si = Nx // 2
            else:
                continue
            # j < Nyr-1 -> same column; j == Nyrs-1 <- source Nyquist column (fft.cu:117-152);
            # destination column Nyr-1 stays zero.
            out[..., i, : Nyr - 1] = freq[..., si, : Nyr - 1]
            out[..., i, Nyrs - 1] = freq[..., si, Nyr - 1]
    return out


def pool_fft(freq, Nx, Ny, scale):
    """fft.cu:975-1002.  Returns (freq', Nx', Ny')."""
    if scale == 1:
        return freq, Nx, Ny
    Nxs, Nys = pooled_size(Nx, Ny, scale)
    return resize(freq, Nx, Ny, Nxs, Nys), Nxs, Nys


# ----------------------------------------------------------------------------------------
# kernel pad / shrink (fft.cu:1018-1064 `kernel_pad`, 1069-1112 `kernel_invpad`,
# 535-565 `shrink_k`, 570-600 `pad_k`)
# ----------------------------------------------------------------------------------------
def _tap_rows(Nk, Nx):
    """tap k -> padded row (k - Nk/2) mod Nx   (fft.cu:544-563 / 1034-1058)."""
    k = np.arange(Nk)
    return np.where(k >= Nk // 2, k - Nk // 2, k + Nx - Nk // 2)


def pad_k(ck, Nx, Ny):
    """fft.cu:570-600 (device) == kernel_pad 1018-1064 (host): [A][B][Nk][Nl] -> [A][B][Nx][Ny]."""
    ck = np.asarray(ck)
    Nk, Nl = ck.shape[-2:]
    out = np.zeros(ck.shape[:-2] + (Nx, Ny), dtype=ck.dtype)
    out[..., _tap_rows(Nk, Nx)[:, None], _tap_rows(Nl, Ny)[None, :]] = ck
    return out


def shrink_k(cpad, Nk, Nl):
    """fft.cu:535-565 (device) == kernel_invpad 1069-1112 (host)."""
    cpad = np.asarray(cpad)
    Nx, Ny = cpad.shape[-2:]
    return cpad[..., _tap_rows(Nk, Nx)[:, None], _tap_rows(Nl, Ny)[None, :]].copy()


def kernel_spectrum(c, Nx, Ny, dtype=np.float64):
    """StoreLoad_cfreq first pass (fft.cu:1150-1152): kernel_pad + kfft."""
    return fft(pad_k(np.asarray(c, dtype=dtype), Nx, Ny), dtype)


def export_cfreq(C, Nx, Ny, Nk, Nl, dtype=np.float64):
    """fft.cu:1166-1172: kfft_inv (C2R * 1/(Nx*Ny), `:948`) + kernel_invpad."""
    return shrink_k(fft_inv(C, Nx, Ny, dtype), Nk, Nl)


# ----------------------------------------------------------------------------------------
# Hadamard channel contraction (fft.cu:162-189 `conv_k`, 1007-1013 `conv_fft`)
# ----------------------------------------------------------------------------------------
def conv_k(X, C, b, Nx, Ny, dtype=np.float64):
    """fft.cu:162-189.  X [dD][Nx][Nyr], C [dM][dD][Nx][Nyr], b [dM] -> O [dM][Nx][Nyr].
    O[m] = sum_d (X[d]/dM) * C[m][d], sequential over d; Re O[m][0][0] += b[m]*Nx*Ny after d==0."""
    ct = _ctype(dtype)
    X = np.asarray(X, dtype=ct)
    C = np.asarray(C, dtype=ct)
    b = np.asarray(b, dtype=dtype)
    dM, dD = C.shape[:2]
    xr = (X.real / dtype(dM)).astype(dtype)
    xi = (X.imag / dtype(dM)).astype(dtype)
    outr = np.zeros((dM,) + X.shape[1:], dtype=dtype)
    outi = np.zeros_like(outr)
    for d in range(dD):
        cr, ci = C[:, d].real, C[:, d].imag
        outr += xr[d] * cr - xi[d] * ci
        outi += xr[d] * ci + xi[d] * cr
        if d == 0:
            outr[:, 0, 0] += b * dtype(Nx) * dtype(Ny)
    return (outr + 1j * outi).astype(ct)


# ----------------------------------------------------------------------------------------
# frequency-space weight gradient (fft.cu:395-475 `gradient_k_io`)
# ----------------------------------------------------------------------------------------
def gradient_k_io(Xin, Xout, O, C, F, b, Nx, Ny, dtype=np.float64):
    """fft.cu:395-475, literal.  Xin = spectrum of the pair's input, Xout = spectrum of the
    expected output, O = spectrum of the autoencoder output, C [dM][dD][..], F [dD][dM][..].
    Returns dc [dM][dD][Nx][Nyr], df [dD][dM][Nx][Nyr], db [dM], dp [dD]."""
    ct = _ctype(dtype)
    Xin, Xout, O = (np.asarray(a, dtype=ct) for a in (Xin, Xout, O))
    C, F = np.asarray(C, dtype=ct), np.asarray(F, dtype=ct)
    b = np.asarray(b, dtype=dtype)
    dM, dD = C.shape[:2]
    norm = dtype(Nx * Ny)                                   # :398
    Norm = dtype(norm * dtype(2 * dM * dD * Nx * Ny))       # :399
    sh = (dM,) + Xin.shape[1:]
    sRR = np.zeros(sh, dtype); sRI = np.zeros(sh, dtype); sIR = np.zeros(sh, dtype); sII = np.zeros(sh, dtype)
    sfR = np.zeros(sh, dtype); sfI = np.zeros(sh, dtype)
    sumb = np.zeros(dM, dtype)
    for d1 in range(dD):                                   # :412-433
        eR = (O[d1].real - Xout[d1].real).astype(dtype)
        eI = (O[d1].imag - Xout[d1].imag).astype(dtype)
        fR, fI = F[d1].real, F[d1].imag                    # [dM][Nx][Nyr]
        sRR += eR * fR; sRI += eR * fI; sIR += eI * fR; sII += eI * fI
        cR, cI = C[:, d1].real, C[:, d1].imag
        sfR += cR * Xin[d1].real - cI * Xin[d1].imag
        sfI += cR * Xin[d1].imag + cI * Xin[d1].real
        sumb += eR[0, 0] * fR[:, 0, 0] + eI[0, 0] * fI[:, 0, 0]
    dc = np.zeros((dM, dD) + Xin.shape[1:], ct)
    df = np.zeros((dD, dM) + Xin.shape[1:], ct)
    b0 = np.zeros(sh, dtype)
    b0[:, 0, 0] = b * norm                                 # :448-450
    for d in range(dD):
        xr, xi = Xin[d].real, Xin[d].imag
        dDR = sRR * xr - sRI * xi + sIR * xi + sII * xr     # :438
        dDI = -sRR * xi - sRI * xr + sIR * xr - sII * xi    # :439
        dc[:, d] = (dDR / Norm) + 1j * (dDI / Norm)
        diffR = (O[d].real - Xout[d].real).astype(dtype)
        diffI = (O[d].imag - Xout[d].imag).astype(dtype)
        fDR = diffR * (sfR + b0) + diffI * sfI              # :454
        fDI = -diffR * sfI + diffI * (sfR + b0)             # :455
        df[d] = (fDR / Norm) + 1j * (fDI / Norm)
    db = (sumb * norm / Norm).astype(dtype)                # :465
    dp = ((O[:, 0, 0].real - Xout[:, 0, 0].real) * norm / Norm).astype(dtype)   # :471
    return dc.astype(ct), df.astype(ct), db, dp


# ----------------------------------------------------------------------------------------
# spectral MSE (fft.cu:480-498 `calc_mse`, 1178-1192 `mse_fft`)
# ----------------------------------------------------------------------------------------
def mse_fft(T, O, dM, dD, Nx, Ny, dtype=np.float64):
    ct = _ctype(dtype)
    T, O = np.asarray(T, dtype=ct), np.asarray(O, dtype=ct)
    Nyr = Ny // 2 + 1
    n = np.full(Nyr, dtype(dD * Nx * Ny), dtype=dtype)
    n[1:Nyr - 1] /= 2                                       # :495
    d = T - O
    cout = ((d.real * d.real + d.imag * d.imag) / n).astype(dtype)
    return dtype(cout.sum(dtype=np.float64)) / dtype(2 * dM * Nx * Ny)   # :1188-1190


# ----------------------------------------------------------------------------------------
# coordinate-space update (fft.cu:605-652 `backprop_d`, 657-704 `backprop_double`,
# 709-753 `gradient_diff`)
# ----------------------------------------------------------------------------------------
ALPHA = 0.9          # fft.cu:608,660
W0, W1 = 1.0, 10.0   # fft.cu:1252


def _clip_step(g, D, dele, dtype):
    """D <- (1-alpha)*del*g/max(10,|g|) + alpha*D   (fft.cu:616)."""
    alpha = dtype(ALPHA)
    ag = np.abs(g)
    den = np.where(dtype(10) < ag, ag, dtype(10))
    return ((dtype(1) - alpha) * dele * g / den + alpha * D).astype(dtype)


def backprop_d(c, f, b, p, dck, dfk, db, dp, Dc, Df, Db, Dp, dele, dtype=np.float64):
    """fft.cu:605-652.  All arrays updated functionally; returns (c,f,b,p,Dc,Df,Db,Dp)."""
    dele = dtype(dele)
    Dc = _clip_step(dck, Dc, dele, dtype); c = (c - Dc).astype(dtype)
    Df = _clip_step(dfk, Df, dele, dtype); f = (f - Df).astype(dtype)
    Db = _clip_step(db, Db, dele, dtype); b = (b - Db).astype(dtype)
    Dp = _clip_step(dp, Dp, dele, dtype); p = (p - Dp).astype(dtype)
    return c, f, b, p, Dc, Df, Db, Dp


def gradient_diff(c, f, b, p, dtype=np.float64):
    """fft.cu:709-753, literal: kernel-distance gradient for the multiobjective mode.
    c [dM][dD][Nk][Nl], f [dD][dM][Nk][Nl]. Pairs need m1!=m AND d1!=d (`:724`).
    Division by zero when two kernels / biases coincide is replicated (inf/nan)."""
    c = np.asarray(c, dtype); f = np.asarray(f, dtype)
    b = np.asarray(b, dtype); p = np.asarray(p, dtype)
    dM, dD = c.shape[:2]
    cd = np.zeros_like(c); fd = np.zeros_like(f)
    ft = np.transpose(f, (1, 0, 2, 3))          # view as [m][d]
    fdt = np.zeros_like(ft)
    with np.errstate(divide="ignore", invalid="ignore"):
        for m in range(dM):
            for d in range(dD):
                sc = np.zeros(c.shape[2:], dtype); sf = np.zeros(c.shape[2:], dtype)
                for m1 in range(dM):
                    for d1 in range(dD):
                        if m1 != m and d1 != d:
                            dcv = c[m, d] - c[m1, d1]
                            dfv = ft[m, d] - ft[m1, d1]
                            sc += dcv / dtype((dcv * dcv).sum())
                            sf += dfv / dtype((dfv * dfv).sum())
                cd[m, d] = sc
                fdt[m, d] = sf
        fd = np.transpose(fdt, (1, 0, 2, 3)).copy()
        bd = np.zeros(dM, dtype); pd = np.zeros(dD, dtype)
        for m in range(dM):
            for m1 in range(dM):
                if m1 != m:
                    bd[m] += dtype(1.0) / (b[m] - b[m1])
        for d in range(dD):
            for d1 in range(dD):
                if d1 != d:
                    pd[d] += dtype(1.0) / (p[d] - p[d1])
    return cd, fd, bd, pd


def gradient_diff_fast(c, f, b, p, block=64, rows=None):
    """fft.cu:709-753 in float64, vectorised: the SAME sums as `gradient_diff` (which follows the source loop by loop) with the
    partner loop as array arithmetic, so that config 5's map counts (64 -> 128: 8192 kernels, 6.7e7 kernel pairs) run in seconds:
        g[i] = sum_{j: m_j != m_i and d_j != d_i} (K_i - K_j) / |K_i - K_j|^2  =  K_i * sum_j w_ij - sum_j w_ij K_j,
    w_ij = mask_ij / |K_i - K_j|^2 with the squared distances formed from the differences themselves (no |a|^2+|b|^2-2ab
    cancellation).  Checked against the literal loop nest in tests/test_oracle_fast.py.  Coinciding kernels give inf/nan as in
    the source (`:724-746`).  Returns (cd, fd, bd, pd) like gradient_diff.  `rows` (indices m*dD + d): only those kernels' sums are
    formed -- cd[rows] and fd^T[rows] as [len(rows)][Nk][Nl] -- for spot checks of tensors too large to do whole."""
    c = np.asarray(c, np.float64); f = np.asarray(f, np.float64)
    b = np.asarray(b, np.float64); p = np.asarray(p, np.float64)
    dM, dD = c.shape[:2]

    def one(K):                                   # K [dM][dD][Nk][Nl] indexed [m][d]
        n = dM * dD
        Kf = K.reshape(n, -1)
        mi = np.repeat(np.arange(dM), dD); di = np.tile(np.arange(dD), dM)
        sel = np.arange(n) if rows is None else np.asarray(rows)
        out = np.zeros((len(sel), Kf.shape[1]))
        with np.errstate(divide="ignore", invalid="ignore"):
            for i0 in range(0, len(sel), block):
                ii = sel[i0:i0 + block]
                diff = Kf[ii, None, :] - Kf[None, :, :]                          # [bi][n][T]
                dist = np.einsum("ijt,ijt->ij", diff, diff)
                mask = (mi[ii, None] != mi[None, :]) & (di[ii, None] != di[None, :])
                w = np.where(mask, 1.0 / np.where(mask, dist, 1.0), 0.0)
                w = np.where(mask & (dist == 0.0), np.inf, w)
                out[i0:i0 + len(ii)] = Kf[ii] * w.sum(axis=1)[:, None] - w @ Kf
        return out.reshape(K.shape if rows is None else (len(sel),) + K.shape[2:])

    cd = one(c)
    fd = one(np.transpose(f, (1, 0, 2, 3)))
    if rows is None:
        fd = np.transpose(fd, (1, 0, 2, 3)).copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        db_ = b[:, None] - b[None, :]; np.fill_diagonal(db_, np.inf)
        dp_ = p[:, None] - p[None, :]; np.fill_diagonal(dp_, np.inf)
        bd = (1.0 / db_).sum(axis=1); pd = (1.0 / dp_).sum(axis=1)
    return cd, fd, bd, pd


def backprop_double(c, f, b, p, dck, dfk, db, dp, Dc, Df, Db, Dp, cd, fd, bd, pd, dele, dtype=np.float64):
    """fft.cu:657-704: g = w0*g_rec - w1*g_diff then the backprop_d rule."""
    w0, w1 = dtype(W0), dtype(W1)
    return backprop_d(c, f, b, p, w0 * dck - w1 * cd, w0 * dfk - w1 * fd, w0 * db - w1 * bd, w0 * dp - w1 * pd,
                      Dc, Df, Db, Dp, dele, dtype)


def backprop(c, f, b, p, dc, df, db, dp, Dc, Df, Db, Dp, Nx, Ny, dele, maxdiff, dtype=np.float64):
    """fft.cu:1197-1291 (host `backprop`): unnormalised C2R of the gradient spectra, shrink,
    update, zero + pad, R2C.  Returns (c,f,b,p,Dc,Df,Db,Dp,C,F)."""
    Nk, Nl = c.shape[-2:]
    dck = shrink_k(c2r_unnorm(dc, Nx, Ny, dtype), Nk, Nl)     # :1219,1225
    dfk = shrink_k(c2r_unnorm(df, Nx, Ny, dtype), Nk, Nl)     # :1220,1226
    if maxdiff:
        cd, fd, bd, pd = gradient_diff(c, f, b, p, dtype)     # :1237
        c, f, b, p, Dc, Df, Db, Dp = backprop_double(c, f, b, p, dck, dfk, db, dp, Dc, Df, Db, Dp,
                                                     cd, fd, bd, pd, dele, dtype)
    else:
        c, f, b, p, Dc, Df, Db, Dp = backprop_d(c, f, b, p, dck, dfk, db, dp, Dc, Df, Db, Dp, dele, dtype)
    C = fft(pad_k(c, Nx, Ny), dtype)                           # :1274-1282
    F = fft(pad_k(f, Nx, Ny), dtype)
    return c, f, b, p, Dc, Df, Db, Dp, C, F


# ----------------------------------------------------------------------------------------
# external entry points (fft.cu:1331-1376 `autoenc_fft`, 1381-1511 `backprop_fft`)
# ----------------------------------------------------------------------------------------
def autoenc_fft(layer0, net_c, net_b, scale, net_cfreq=None, dtype=np.float64):
    """fft.cu:1331-1376 with fft_l=1 semantics (every intermediate is also returned in
    coordinate space).  net_c = encoders 0..L-1 then decoders mirrored; scale = [+s.., -s..].
    Returns (layers, net_cfreq, spectra) where layers follows autoencoder.cpp:110-114
    ([in, Pin0, hC0, ..., PhC0, out0]) and spectra is the list of the same tensors in
    frequency space."""
    x = np.asarray(layer0, dtype=dtype)
    dD, Nx, Ny = x.shape
    freq = fft(x, dtype)
    layers, spectra = [x], [freq]
    if net_cfreq is None:
        net_cfreq = []
    N = len(net_c)
    for n in range(N):
        dM = len(net_c[n])
        if n < N // 2:
            freq, Nx, Ny = pool_fft(freq, Nx, Ny, scale[n])          # :1346
            layers.append(fft_inv(freq, Nx, Ny, dtype)); spectra.append(freq)
        if len(net_cfreq) <= n:                                       # :1148-1158
            net_cfreq.append(kernel_spectrum(net_c[n], Nx, Ny, dtype))
        ofreq = conv_k(freq, net_cfreq[n], net_b[n], Nx, Ny, dtype)   # :1356
        layers.append(fft_inv(ofreq, Nx, Ny, dtype)); spectra.append(ofreq)
        if n >= N // 2:
            ofreq, Nx, Ny = pool_fft(ofreq, Nx, Ny, scale[n])         # :1360
            layers.append(fft_inv(ofreq, Nx, Ny, dtype)); spectra.append(ofreq)
        freq = ofreq
    return layers, net_cfreq, spectra


def backprop_fft(in_s, expout, out_s, C, c, F, f, b, p, del0, maxdiff=0, n_iter=100, dtype=np.float64,
                 spectra=None):
    """fft.cu:1381-1511.  in_s/expout/out_s [dD][Nx][Ny]; C [dM][dD][Nx][Nyr]; F [dD][dM][..].
    Returns dict(c,f,b,p,C,F,mse=[initial, after iter 0, ...]).  `n_iter` is 100 in the
    reference (`:1446`); `del = 0.1*del0` (`:1445`); momentum buffers start at zero (`:1420-1423`).
    `spectra=(X,T,O)` skips the three R2C's (used by the spectrum-resident batch path)."""
    c = np.asarray(c, dtype); f = np.asarray(f, dtype)
    b = np.asarray(b, dtype); p = np.asarray(p, dtype)
    dM, dD, Nk, Nl = c.shape
    if spectra is None:
        Nx, Ny = np.asarray(in_s).shape[-2:]
        X, T, O = fft(in_s, dtype), fft(expout, dtype), fft(out_s, dtype)   # :1430-1432
    else:
        X, T, O = spectra
        Nx = X.shape[-2]; Ny = (X.shape[-1] - 1) * 2
    C = np.asarray(C, _ctype(dtype)); F = np.asarray(F, _ctype(dtype))
    Dc = np.zeros_like(c); Df = np.zeros_like(f); Db = np.zeros_like(b); Dp = np.zeros_like(p)
    mses = [mse_fft(T, O, dM, dD, Nx, Ny, dtype)]                            # :1440
    dele = dtype(0.1) * dtype(del0)                                          # :1445
    for _ in range(n_iter):
        dc, df, db, dp = gradient_k_io(X, T, O, C, F, b, Nx, Ny, dtype)      # :1454
        c, f, b, p, Dc, Df, Db, Dp, C, F = backprop(c, f, b, p, dc, df, db, dp, Dc, Df, Db, Dp,
                                                    Nx, Ny, dele, maxdiff, dtype)   # :1456
        H = conv_k(X, C, b, Nx, Ny, dtype)                                   # :1460
        O = conv_k(H, F, p, Nx, Ny, dtype)                                   # :1461
        mses.append(mse_fft(T, O, dM, dD, Nx, Ny, dtype))                    # :1463
    return dict(c=c, f=f, b=b, p=p, C=C, F=F, H=H if n_iter else None, O=O, mse=mses)


# ----------------------------------------------------------------------------------------
# build-defined data-parallel batch step (no reference counterpart):
# per-frame gradient spectra -> C2R -> shrink are linear, so the shrunk gradients are summed
# over the B frames, divided by B, and fed to ONE backprop_d update.  B=1 == reference.
# ----------------------------------------------------------------------------------------
def batch_grad(Xs, Ts, Os, C, F, b, Nk, Nl, dtype=np.float64):
    """Mean over frames of the shrunk coordinate-space gradients (dck, dfk, db, dp)."""
    B = len(Xs)
    Nx = Xs[0].shape[-2]; Ny = (Xs[0].shape[-1] - 1) * 2
    acc = None
    for X, T, O in zip(Xs, Ts, Os):
        dc, df, db, dp = gradient_k_io(X, T, O, C, F, b, Nx, Ny, dtype)
        g = [shrink_k(c2r_unnorm(dc, Nx, Ny, dtype), Nk, Nl),
             shrink_k(c2r_unnorm(df, Nx, Ny, dtype), Nk, Nl), db, dp]
        acc = g if acc is None else [a + gi for a, gi in zip(acc, g)]
    return [(a / dtype(B)).astype(dtype) for a in acc]


def batch_train_iter(Xs, Ts, Os, C, F, c, f, b, p, mom, dele, maxdiff=0, dtype=np.float64):
    """One iteration of the backprop_fft loop body over a batch (gradient mean -> update ->
    new spectra -> per-frame re-forward -> mean MSE).  mom = (Dc,Df,Db,Dp)."""
    dM, dD, Nk, Nl = c.shape
    Nx = Xs[0].shape[-2]; Ny = (Xs[0].shape[-1] - 1) * 2
    dck, dfk, db, dp = batch_grad(Xs, Ts, Os, C, F, b, Nk, Nl, dtype)
    Dc, Df, Db, Dp = mom
    if maxdiff:
        cd, fd, bd, pd = gradient_diff(c, f, b, p, dtype)
        c, f, b, p, Dc, Df, Db, Dp = backprop_double(c, f, b, p, dck, dfk, db, dp, Dc, Df, Db, Dp,
                                                     cd, fd, bd, pd, dtype(dele), dtype)
    else:
        c, f, b, p, Dc, Df, Db, Dp = backprop_d(c, f, b, p, dck, dfk, db, dp, Dc, Df, Db, Dp, dtype(dele), dtype)
    C = fft(pad_k(c, Nx, Ny), dtype); F = fft(pad_k(f, Nx, Ny), dtype)
    Hs = [conv_k(X, C, b, Nx, Ny, dtype) for X in Xs]
    Os2 = [conv_k(H, F, p, Nx, Ny, dtype) for H in Hs]
    mse = dtype(np.mean([mse_fft(T, O, dM, dD, Nx, Ny, dtype) for T, O in zip(Ts, Os2)]))
    return dict(c=c, f=f, b=b, p=p, C=C, F=F, mom=(Dc, Df, Db, Dp), Hs=Hs, Os=Os2, mse=mse,
                grads=(dck, dfk, db, dp))


# ----------------------------------------------------------------------------------------
# build-defined tied-weight update for FFT mode (the reference has tied weights
# only in spatial mode, backproplib.cu:521-644, whose rule is mirrored): g = (g_c[m][d] + g_f[d][m]) / 2
# (i.e. the sum with Norm doubled, :533), biases' gradients halved likewise, c <- c - D, f[d][m] <- c[m][d].
# ----------------------------------------------------------------------------------------
def backprop_sym(c, f, b, p, dck, dfk, db, dp, Dc, Df, Db, Dp, dele, cd=None, fd=None, bd=None, pd=None, dtype=np.float64):
    dele = dtype(dele)
    g = 0.5 * (dck + np.transpose(dfk, (1, 0, 2, 3)))
    gb, gp = 0.5 * db, 0.5 * dp
    if cd is not None:
        g = dtype(W0) * g - dtype(W1) * 0.5 * (cd + np.transpose(fd, (1, 0, 2, 3)))
        gb = dtype(W0) * gb - dtype(W1) * bd
        gp = dtype(W0) * gp - dtype(W1) * pd
    Dc = _clip_step(g, Dc, dele, dtype); c = (c - Dc).astype(dtype)
    f = np.transpose(c, (1, 0, 2, 3)).copy()
    Db = _clip_step(gb, Db, dele, dtype); b = (b - Db).astype(dtype)
    Dp = _clip_step(gp, Dp, dele, dtype); p = (p - Dp).astype(dtype)
    return c, f, b, p, Dc, Df, Db, Dp


# ----------------------------------------------------------------------------------------
# the build-defined training step over a whole network (SURVEY 8d "one frame fwd+bwd", 8e): autoenc_fft of every frame, then for
# EVERY pair one backprop_fft loop-body iteration on the spectra of that forward (batch-mean gradients), momentum carried from
# step to step (aefft_net_step_grad / _apply).  Used by the trajectory tests; one call = one step.
# ----------------------------------------------------------------------------------------
def net_step(xs, ws, moms, scale, del0, maxdiff=0, sym=0, dtype=np.float64, fast_diff=True):
    """xs [B][D][Nx][Ny]; ws = [(c, b, f, p)] per pair; moms = [(Dc, Df, Db, Dp)] per pair (None: zeros).  Returns
    (ws', moms', mse per pair [post-update, fft.cu:1463], recon [B][D][Nx][Ny])."""
    L = len(ws)
    net_c = [np.asarray(w[0], dtype) for w in ws] + [np.asarray(w[2], dtype) for w in ws[::-1]]
    net_b = [np.asarray(w[1], dtype) for w in ws] + [np.asarray(w[3], dtype) for w in ws[::-1]]
    cf = None
    sp = []
    for x in xs:
        layers, cf, spec = autoenc_fft(np.asarray(x, dtype), net_c, net_b, [scale] * L + [-scale] * L, net_cfreq=cf, dtype=dtype)
        sp.append((layers, spec))
    recon = np.stack([q[0][-1] for q in sp])
    dele = dtype(0.1) * dtype(del0)
    out_w, out_m, mses = [], [], []
    for l in range(L):
        c, b, f, p = (np.asarray(a, dtype) for a in ws[l])
        mom = moms[l] if moms is not None and moms[l] is not None else tuple(np.zeros_like(a) for a in (c, f, b, p))
        Xs = [q[1][2 * l + 1] for q in sp]; Os = [q[1][4 * L - 1 - 2 * l] for q in sp]
        dM, dD, Nk, Nl = c.shape
        Nx = Xs[0].shape[-2]; Ny = (Xs[0].shape[-1] - 1) * 2
        if not sym:
            r = batch_train_iter(Xs, Xs, Os, cf[l], cf[2 * L - 1 - l], c, f, b, p, mom, dele, maxdiff, dtype)
            out_w.append((r["c"], r["b"], r["f"], r["p"])); out_m.append(r["mom"]); mses.append(r["mse"])
            continue
        dck, dfk, db, dp = batch_grad(Xs, Xs, Os, cf[l], cf[2 * L - 1 - l], b, Nk, Nl, dtype)
        extra = ()
        if maxdiff:
            extra = tuple(np.asarray(a, dtype) for a in (gradient_diff_fast(c, f, b, p) if fast_diff else gradient_diff(c, f, b, p, dtype)))
        c2, f2, b2, p2, Dc, Df, Db, Dp = backprop_sym(c, f, b, p, dck, dfk, db, dp, *mom, dele, *extra, dtype=dtype)
        C = fft(pad_k(c2, Nx, Ny), dtype); F = fft(pad_k(f2, Nx, Ny), dtype)
        Os2 = [conv_k(conv_k(X, C, b2, Nx, Ny, dtype), F, p2, Nx, Ny, dtype) for X in Xs]
        mses.append(dtype(np.mean([mse_fft(X, O, dM, dD, Nx, Ny, dtype) for X, O in zip(Xs, Os2)])))
        out_w.append((c2, b2, f2, p2)); out_m.append((Dc, Df, Db, Dp))
    return out_w, out_m, mses, recon
