"""Generates the committed golden vectors.  Run in the build container (needs /root/reference
for the CPU-path vectors, which come from the reference's own compiled code, oracle/_ref):

    python tests/golden/make_golden.py

cpu_path.npz  : inputs + outputs of the compiled reference Conv/Conv/backprop
                (BASELINE config 1: 128x128 gray, 4 maps, 3x3; plus a small 5x5 case).
fft_path.npz  : inputs + float64 outputs of oracle/np_ref.py for the FFT-mode path
                (autoenc_fft + backprop_fft bursts, small shapes).  The reference's CUDA
                path cannot run here, so these vectors pin the HIP path to the restatement,
                and the restatement is cross-pinned to the compiled CPU reference in
                tests/test_oracle_crosspin.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import cpu          # noqa: E402
import np_ref as R  # noqa: E402


def cpu_case(seed_x, seed_w, dD, dM, N, Nk):
    x = np.floor(np.random.default_rng(seed_x).uniform(0, 256, (dD, N, N))).astype(np.float32)
    rw = np.random.default_rng(seed_w)
    c = rw.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32)
    f = rw.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    b = rw.uniform(-1, 1, dM).astype(np.float32)
    p = rw.uniform(-1, 1, dD).astype(np.float32)
    return x, c, b, f, p


def make_cpu():
    ref = cpu.reference()
    assert ref is not None, "needs /root/reference (oracle/_ref)"
    out = {}
    for tag, args in (("cfg1", (1, 2, 1, 4, 128, 3)), ("k5", (3, 4, 2, 3, 20, 5))):
        x, c, b, f, p = cpu_case(*args)
        pin = ref.pool(x, x.shape, 1)
        h = ref.conv(pin, c, b)
        o = ref.conv(h, f, p)
        c2, b2, f2, p2 = ref.backprop(pin, o, h, c, b, f, p, 0.2)
        out.update({f"{tag}_{k}": v for k, v in dict(x=x, c=c, b=b, f=f, p=p, h=h, o=o, c2=c2, b2=b2, f2=f2, p2=p2).items()})
    np.savez_compressed(os.path.join(HERE, "cpu_path.npz"), **out)


def net_weights(rng, D, maps, Nk, rmax):
    """encoders 0..L-1 then mirrored decoders (autoencoder.cpp:115-118,414-418)."""
    enc, dec, eb, db = [], [], [], []
    dD = D
    for dM in maps:
        enc.append(rng.uniform(-rmax, rmax, (dM, dD, Nk, Nk))); eb.append(rng.uniform(-rmax, rmax, dM))
        dec.append(rng.uniform(-rmax, rmax, (dD, dM, Nk, Nk))); db.append(rng.uniform(-rmax, rmax, dD))
        dD = dM
    return enc + dec[::-1], eb + db[::-1]


def frame(seed, D, N):
    """SURVEY 8d synthetic frame: floor(U[0,256)) + smooth low-frequency component."""
    rng = np.random.default_rng(seed)
    x = np.floor(rng.uniform(0, 256, (D, N, N)))
    i = np.arange(N)[:, None] / N; j = np.arange(N)[None, :] / N
    for d in range(D):
        x[d] = 0.5 * x[d] + 64 * (1 + np.sin(2 * np.pi * (i * (d + 1) + 0.3))) * (1 + np.cos(2 * np.pi * j * 2)) / 2
    return np.floor(x)


def make_fft():
    out = {}
    # case A: 1 pair, no pooling, 16x16, 3->4, 5x5, 3 iterations, with and without multiobjective
    # case B: 2 pairs, s=2, 32x32, 3->4->6, 3x3: forward layers + a burst on each pair
    for tag, (D, N, maps, Nk, s, rmax) in dict(A=(3, 16, [4], 5, 1, 1.0), B=(3, 32, [4, 6], 3, 2, 1.0)).items():
        rng = np.random.default_rng(100 + ord(tag))
        net_c, net_b = net_weights(rng, D, maps, Nk, rmax)
        L = len(maps)
        scale = [s] * L + [-s] * L
        x = frame(1000, D, N)
        layers, cfreq, spectra = R.autoenc_fft(x, net_c, net_b, scale)
        out[f"{tag}_x"] = x.astype(np.float32)
        for n, (c, b) in enumerate(zip(net_c, net_b)):
            out[f"{tag}_c{n}"] = c.astype(np.float32); out[f"{tag}_b{n}"] = b.astype(np.float32)
        for l, lay in enumerate(layers):
            out[f"{tag}_layer{l}"] = lay
        for n_l in range(L):
            Nn = len(net_c)
            in_s = layers[2 * n_l + 1]; out_s = layers[len(layers) - 2 - 2 * n_l]
            for md in (0, 1):
                r = R.backprop_fft(in_s, in_s, out_s, cfreq[n_l], net_c[n_l], cfreq[Nn - 1 - n_l], net_c[Nn - 1 - n_l],
                                   net_b[n_l], net_b[Nn - 1 - n_l], 0.2, maxdiff=md, n_iter=3)
                for k in ("c", "f", "b", "p"):
                    out[f"{tag}_burst{n_l}_md{md}_{k}"] = r[k]
                out[f"{tag}_burst{n_l}_md{md}_mse"] = np.array(r["mse"])
    np.savez_compressed(os.path.join(HERE, "fft_path.npz"), **out)


if __name__ == "__main__":
    make_cpu()
    make_fft()
    for fn in ("cpu_path.npz", "fft_path.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")
