"""TEST INFRASTRUCTURE ONLY -- literal, per-weight-element restatement of the reference's spatial-mode GPU gradient
(/root/reference/source/backproplib.cu:186-288 `gradient_CFBP` / `gradient_CF` and the host loop `:363-417` that launches one of
them per weight element (m, d, k, l) and reduces the per-pixel buffers).  Imported only by tests/.

Why it exists: oracle/np_spatial.py computes the same gradients through a two-stage re-association (back-convolution, then
correlation) -- the SAME re-association the HIP kernels use -- so a shared misreading of the geometry would pass unnoticed.
This file follows the CUDA source loop by loop instead (vectorised over the pixel grid only, which is what one launch covers).

`compat=False`: the terms SURVEY Appendix B-11 calls bugs follow the CPU reference (netlib.cpp:425-430), i.e. what
np_spatial.py and the HIP kernels compute by default.
`compat=True`: the CUDA source exactly as written --
    * `dDdB2 = ...` instead of `+=` (`:220`): only the LAST d1 contributes to the encoder-bias gradient;
    * the hidden layer is read at flat index (i-ik)*Nx + (j-il) in gradient_CFBP (`:226`, row stride Nx, not Ny) and at
      (i-ik)*Nx + (j-ik) in gradient_CF (`:283`, row stride Nx AND the column shifted by ik); a flat index that leaves the hin
      buffer is undefined behaviour in the reference and reads 0 here;
    * pixels whose shifted position is out of range keep the dDdF value of the PREVIOUS launch (`:225-227`, `:282-284`: the
      buffer is only written inside the range test; it starts zeroed, `:335`).
Parity status: unpinned by reference-held vectors (CUDA cannot run here); pinned to oracle/_ref under CPU geometry through
np_spatial (tests/test_oracle_crosspin.py).
"""
import numpy as np


def _geom(Nk, Nl):
    return ((Nk - 1) // 2 - 1) // 2, ((Nl - 1) // 2 - 1) // 2          # backproplib.cu:301-302


def gradients_literal(x, out, hin, f, compat=False, tied=False, dtype=np.float64):
    """x, out [dD][Nx][Ny], hin [dM][Nx][Ny], f [dD][dM][Nk][Nl] -> gc [dM][dD][Nk][Nl], gf [dD][dM][Nk][Nl], gb [dM], gp [dD]."""
    x = np.asarray(x, dtype); out = np.asarray(out, dtype); hin = np.asarray(hin, dtype); f = np.asarray(f, dtype)
    dD, dM, Nk, Nl = f.shape
    Nx, Ny = x.shape[-2:]
    ak, al = _geom(Nk, Nl)
    Norm = dtype(dD * dM * Nk * Nl * Nx * Ny) * (2 if tied else 1)       # :303 / :533
    I, J = np.meshgrid(np.arange(Nx), np.arange(Ny), indexing="ij")      # one launch covers the pixel grid (:371-372)
    hflat = hin.reshape(-1)
    gc = np.zeros((dM, dD, Nk, Nl), dtype); gf = np.zeros((dD, dM, Nk, Nl), dtype)
    gb = np.zeros(dM, dtype); gp = np.zeros(dD, dtype)
    dDdF_buf = np.zeros((Nx, Ny), dtype)                                 # thrust::device_vector: zero-initialised (:335)

    def at(a, ii, jj):
        """a[ii][jj] where in range, else 0 (the guarded reads of :208,213)"""
        ok = (ii >= 0) & (ii < Nx) & (jj >= 0) & (jj < Ny)
        return np.where(ok, a[np.clip(ii, 0, Nx - 1), np.clip(jj, 0, Ny - 1)], 0), ok

    for m in range(dM):                                                  # host loop :363-370
        for d in range(dD):
            for k in range(Nk):
                ik = -2 * ak - 1 + k
                for l in range(Nl):
                    il = -2 * al - 1 + l
                    first = (k == 0 and l == 0)                          # gradient_CFBP, else gradient_CF (:373)
                    dDdC2 = np.zeros((Nx, Ny), dtype); dDdB2 = np.zeros((Nx, Ny), dtype); dDdP = np.zeros((Nx, Ny), dtype)
                    for d1 in range(dD):
                        dDdB1 = np.zeros((Nx, Ny), dtype); dDdC1 = np.zeros((Nx, Ny), dtype)
                        for k1 in range(Nk):
                            ik1 = -2 * ak - 1 + k1
                            for l1 in range(Nl):
                                il1 = -2 * al - 1 + l1
                                ok1 = (I - ik1 >= 0) & (I - ik1 < Nx) & (J - il1 >= 0) & (J - il1 < Ny)       # :208 / :257
                                prod = np.where(ok1, f[d1, m, k1, l1], 0)                                   # act1_d == 1 (:62-67)
                                dDdB1 += prod
                                xin, ok2 = at(x[d], I - ik1 - ik, J - il1 - il)                             # :212-213
                                dDdC1 += prod * xin
                        sum0 = out[d1] - x[d1]                                                              # :217 (act1_d == 1)
                        dDdC2 += sum0 * dDdC1 / Norm
                        if compat:
                            dDdB2 = sum0 * dDdB1 / Norm                                                     # :220 '=' as written
                        else:
                            dDdB2 += sum0 * dDdB1 / Norm
                        if d1 == d:
                            okf = (I - ik >= 0) & (I - ik < Nx) & (J - il >= 0) & (J - il < Ny)             # :225 / :282
                            if compat:
                                col = (J - il) if first else (J - ik)                                      # :226 vs :283
                                idx = m * Nx * Ny + (I - ik) * Nx + col
                                inb = (idx >= 0) & (idx < hflat.size)
                                hv = np.where(inb, hflat[np.clip(idx, 0, hflat.size - 1)], 0)
                                dDdF_buf = np.where(okf, sum0 * hv / Norm, dDdF_buf)                        # stale outside the range
                            else:
                                hv, _ = at(hin[m], I - ik, J - il)
                                dDdF_buf = np.where(okf, sum0 * hv / Norm, 0)
                            dDdP = sum0 / Norm                                                              # :227
                    gc[m, d, k, l] = dDdC2.sum()                                                            # thrust::reduce (:384)
                    gf[d, m, k, l] = dDdF_buf.sum()                                                         # :385
                    if first:
                        if d == 0:
                            gb[m] = dDdB2.sum()                                                             # :398-403
                        if m == 0:
                            gp[d] = dDdP.sum()                                                              # :405-411
    if tied:
        pass   # backprop_gpu_cc sums the two gradients on the host (:602-607); callers add gc + gf^T
    return gc, gf, gb, gp
