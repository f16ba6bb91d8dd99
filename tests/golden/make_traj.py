"""Generates tests/golden/traj300.npz: the per-pair post-update MSE of 300 training steps in the regime bench.py times
(del0 = 0.2, autoencoder.cpp:87; weights U(-3,3), New_Layer_Param.txt:5; 4 pairs 3->8->16->32->64, 5x5, pool 2 per layer;
video-like frames as bench.py's synth_frames) at reduced planes (128x128, B = 2), from oracle/np_ref.net_step in float64 (the
master) and in float32 (the oracle's own replay of the reference's arithmetic).  ~1 minute of CPU:

    python tests/golden/make_traj.py

Inputs are NOT stored: `case()` below re-creates them from the seed, and tests/test_gpu_round4.py + tests/test_oracle_fast.py
import it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))

CFG = dict(D=3, N=128, maps=[8, 16, 32, 64], Nk=5, s=2, B=2, del0=0.2, rmax=3.0, seed=4242, steps=300)


def case():
    """(frames [B][D][N][N] float64, weights [(c, b, f, p)] float32-representable float64)"""
    g = CFG
    rng = np.random.default_rng(g["seed"])
    q = lambda a: a.astype(np.float32).astype(np.float64)
    ws, dD = [], g["D"]
    for dM in g["maps"]:
        ws.append((q(rng.uniform(-g["rmax"], g["rmax"], (dM, dD, g["Nk"], g["Nk"]))), q(rng.uniform(-g["rmax"], g["rmax"], dM)),
                   q(rng.uniform(-g["rmax"], g["rmax"], (dD, dM, g["Nk"], g["Nk"]))), q(rng.uniform(-g["rmax"], g["rmax"], dD))))
        dD = dM
    N = g["N"]
    i = np.arange(N)[:, None] / N
    j = np.arange(N)[None, :] / N
    xs = np.empty((g["B"], g["D"], N, N))
    for b in range(g["B"]):
        for d in range(g["D"]):
            smooth = 64 * (1 + np.sin(2 * np.pi * (i * (d + 1) + 0.3))) * (1 + np.cos(2 * np.pi * j * 2)) / 2
            xs[b, d] = np.floor(0.5 * np.floor(rng.uniform(0, 256, (N, N))) + smooth)
    return xs, ws


def run(dtype, steps):
    import np_ref as R
    xs, ws = case()
    w = [tuple(a.astype(dtype) for a in x) for x in ws]
    m, seq = None, []
    for _ in range(steps):
        w, m, mse, _ = R.net_step(xs.astype(dtype), w, m, CFG["s"], CFG["del0"], dtype=dtype)
        seq.append([float(v) for v in mse])
    return np.array(seq)


if __name__ == "__main__":
    m64 = run(np.float64, CFG["steps"])
    m32 = run(np.float32, CFG["steps"])
    np.savez_compressed(os.path.join(HERE, "traj300.npz"), mse64=m64, mse32=m32)
    rel = np.abs(m32 - m64) / np.abs(m64)
    for l in range(m64.shape[1]):
        bad = np.nonzero(rel[:, l] > 1e-3)[0]
        print("pair", l, "float32 replay leaves the master (1e-3) at step", int(bad[0]) if bad.size else None)
