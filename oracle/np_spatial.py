"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's spatial-mode GPU path
(/root/reference/source/backproplib.cu).  Imported only by tests/, smoke() and bench.py's
cpu_baseline leg.

Parity status: the CUDA file cannot be built here and the reference ships no vectors for it
("parity unpinned" by reference-held fixtures).  It is cross-pinned to the reference's compiled
CPU code (oracle/_ref) with `lo=1, cpu_geom=True`, where both compute the same sums
(tests/test_oracle_crosspin.py::test_spatial_*).

Semantics (SURVEY Appendix B-10..13,15): GPU tap offset ik = -2*ak-1+k with ak=((Nk-1)/2-1)/2
(backproplib.cu:123-124), range test '>=0' (`:95`), input divided by dM on the host (`:134`),
gradient from a SNAPSHOT of f (`:322-334`).  The index / stale-buffer bugs of gradient_CF
(`:283`, `:225-227`) and the `dDdB2=` of gradient_CFBP (`:220`) are not replicated: those terms
follow the CPU reference (netlib.cpp:425-430), as SURVEY B-11 prescribes.
"""
import numpy as np


def geom(Nk, Nl, cpu_geom=False):
    if cpu_geom:
        return (Nk - 1) // 2 - 1, (Nl - 1) // 2 - 1      # netlib.cpp:325-326
    return ((Nk - 1) // 2 - 1) // 2, ((Nl - 1) // 2 - 1) // 2   # backproplib.cu:123-124


def _shift(a, di, dj, lo):
    """out[i][j] = a[i-di][j-dj] where (i-di, j-dj) in [lo, N), else 0."""
    Nx, Ny = a.shape[-2:]
    out = np.zeros_like(a)
    i0, i1 = max(0, lo + di), min(Nx, Nx + di)
    j0, j1 = max(0, lo + dj), min(Ny, Ny + dj)
    if i0 < i1 and j0 < j1:
        out[..., i0:i1, j0:j1] = a[..., i0 - di:i1 - di, j0 - dj:j1 - dj]
    return out


def conv(x, c, b, cpu_semantics=False, dtype=np.float64):
    """backproplib.cu:70-111,114-182 (Conv_gpu) or, with cpu_semantics, netlib.cpp:318-358 (Conv).
    x [dD][Nx][Ny], c [dM][dD][Nk][Nl] -> [dM][Nx][Ny]."""
    x = np.asarray(x, dtype); c = np.asarray(c, dtype); b = np.asarray(b, dtype)
    dM, dD, Nk, Nl = c.shape
    ak, al = geom(Nk, Nl, cpu_semantics)
    lo = 1 if cpu_semantics else 0
    xin = x if cpu_semantics else (x / dtype(dM)).astype(dtype)
    out = np.zeros((dM,) + x.shape[1:], dtype)
    for d in range(dD):
        for k in range(Nk):
            for l in range(Nl):
                sh = _shift(xin[d], -2 * ak - 1 + k, -2 * al - 1 + l, lo)
                out += c[:, d, k, l][:, None, None] * sh[None]
    return (out + b[:, None, None]).astype(dtype)


def gradients(x, out, hin, f, tied=False, lo=0, cpu_geom=False, dtype=np.float64):
    """backproplib.cu:186-288 (gradient_CFBP/CF) summed as the host does (`:387-408`), as one batch.
    Returns gc [dM][dD][Nk][Nl], gf [dD][dM][Nk][Nl], gb [dM], gp [dD]."""
    x = np.asarray(x, dtype); out = np.asarray(out, dtype); hin = np.asarray(hin, dtype); f = np.asarray(f, dtype)
    dD, dM, Nk, Nl = f.shape
    Nx, Ny = x.shape[-2:]
    ak, al = geom(Nk, Nl, cpu_geom)
    Norm = dtype(dD * dM * Nk * Nl * Nx * Ny) * (2 if tied else 1)
    s0 = out - x
    # back-conv g[m][i'][j'] = sum_{d1,k1,l1} s0[d1][i'+ik1][j'+il1] f[d1][m][k1][l1], i' >= lo
    g = np.zeros((dM, Nx, Ny), dtype)
    for d1 in range(dD):
        for k1 in range(Nk):
            for l1 in range(Nl):
                ik1, il1 = -2 * ak - 1 + k1, -2 * al - 1 + l1
                sh = _shift(s0[d1], -ik1, -il1, 0)         # sh[i'][j'] = s0[i'+ik1][j'+il1]
                g += f[d1, :, k1, l1][:, None, None] * sh[None]
    g[:, :lo, :] = 0; g[:, :, :lo] = 0
    gc = np.zeros((dM, dD, Nk, Nl), dtype); gf = np.zeros((dD, dM, Nk, Nl), dtype)
    for k in range(Nk):
        for l in range(Nl):
            ik, il = -2 * ak - 1 + k, -2 * al - 1 + l
            xs = _shift(x, ik, il, lo)                      # xs[d][i'][j'] = x[d][i'-ik][j'-il]
            gc[:, :, k, l] = np.einsum("mij,dij->md", g, xs)
            hs = _shift(hin, ik, il, lo)
            gf[:, :, k, l] = np.einsum("dij,mij->dm", s0, hs)
    gb = g.sum(axis=(1, 2)); gp = s0.sum(axis=(1, 2))
    return (gc / Norm).astype(dtype), (gf / Norm).astype(dtype), (gb / Norm).astype(dtype), (gp / Norm).astype(dtype)


def _step(g, d, delmax, alpha, dtype):
    ag = np.abs(g)
    return ((1 - dtype(alpha)) * dtype(delmax) * g / np.where(10 < ag, ag, 10) + dtype(alpha) * d).astype(dtype)


def backprop_gpu(x, out, hin, c, b, f, p, dc, db, df, dp, delmax, alpha, tied=False, dtype=np.float64, B_mean=None):
    """backproplib.cu:291-418 (tied=False) / 521-644 (tied=True).  Returns the updated
    (c,b,f,p, dc,db,df,dp, ddc,ddb,ddf,ddp); adapt_rate is inert except for recording the gradient."""
    if B_mean is None:
        gc, gf, gb, gp = gradients(x, out, hin, f, tied, 0, False, dtype)
    else:
        gs = [gradients(xx, oo, hh, f, tied, 0, False, dtype) for xx, oo, hh in zip(x, out, hin)]
        gc, gf, gb, gp = (sum(t) / len(gs) for t in zip(*gs))
    c = np.asarray(c, dtype).copy(); f = np.asarray(f, dtype).copy()
    if tied:
        g = gc + np.transpose(gf, (1, 0, 2, 3))
        dc = _step(g, dc, delmax, alpha, dtype); c = c - dc
        f = np.transpose(c, (1, 0, 2, 3)).copy()
        ddc, ddf = g, None
    else:
        dc = _step(gc, dc, delmax, alpha, dtype); c = c - dc
        df = _step(gf, df, delmax, alpha, dtype); f = f - df
        ddc, ddf = gc, gf
    db = _step(gb, db, delmax, alpha, dtype); b = np.asarray(b, dtype) - db
    dp = _step(gp, dp, delmax, alpha, dtype); p = np.asarray(p, dtype) - dp
    return c, b, f, p, dc, db, df, dp, ddc, gb, ddf, gp
