"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank reduces ITS shard of the frames to the
packed kernel-support gradient buffer (with the oracle standing in for the device step), the buffer is
all-reduced, and the replicated update must equal the single-process update over the whole batch."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _problem():
    _setup()
    import np_ref as R
    rng = np.random.default_rng(42)
    D, N, maps, Nk = 2, 16, [3, 4], 3
    B = 4
    xs = np.floor(rng.uniform(0, 256, (B, D, N, N)))
    dims, ws, dD = [], [], D
    for dM in maps:
        c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk))
        ws.append((c, rng.uniform(-1, 1, dM), f, rng.uniform(-1, 1, dD)))
        dims.append(dict(dM=dM, dD=dD, Nk=Nk, Nl=Nk)); dD = dM
    net_c = [w[0] for w in ws] + [w[2] for w in ws[::-1]]
    net_b = [w[1] for w in ws] + [w[3] for w in ws[::-1]]
    L = len(maps)
    sp = [R.autoenc_fft(x, net_c, net_b, [1] * L + [-1] * L) for x in xs]
    return R, xs, ws, dims, sp, L


def _local_grads(R, sp, ws, dims, L, idx):
    """mean over the frames `idx` of the shrunk gradients of every pair (what aefft_net_step_grad leaves in the buffer)."""
    out = []
    for l in range(L):
        c, b, f, p = ws[l]
        Xs = [sp[i][2][2 * l + 1] for i in idx]; Os = [sp[i][2][4 * L - 1 - 2 * l] for i in idx]
        out.append(R.batch_grad(Xs, Xs, Os, sp[0][1][l], sp[0][1][2 * L - 1 - l], b, dims[l]["Nk"], dims[l]["Nl"]))
    return out


def _update(R, ws, grads, dele):
    res = []
    for (c, b, f, p), (dck, dfk, db, dp) in zip(ws, grads):
        z = lambda a: np.zeros_like(a)
        res.append(R.backprop_d(c, f, b, p, dck, dfk, db, dp, z(c), z(f), z(b), z(p), dele)[:4])
    return res


def _worker(rank, world, port, q):
    _setup()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("autoencoder-fft_amd.dp")
    R, xs, ws, dims, sp, L = _problem()
    shard = list(range(rank * len(xs) // world, (rank + 1) * len(xs) // world))
    # the packed buffer as the library exposes it: gradients, then one post-update MSE per pair of the previous step on this rank
    local_mse = np.array([10.0 * rank + l + 1 for l in range(L)], np.float32)
    buf = torch.from_numpy(np.concatenate([dp.pack_grads(_local_grads(R, sp, ws, dims, L, shard), dims), local_mse]))
    scale = dp.allreduce_sum_(buf)
    grads = [tuple(scale * a.astype(np.float64) for a in g) for g in dp.unpack_grads(buf.numpy(), dims)]
    new = _update(R, ws, grads, 0.02)
    tail = buf.numpy()[dp.mse_tail_slice(dims)] * scale            # global-batch mean of the MSEs (the MSE rides in the same all-reduce)
    q.put((rank, [[a.copy() for a in t] for t in new], tail.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process_batch():
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    got = {r[0]: r[1] for r in res}
    for r in res:                                                          # every rank reads the same global mean: (l+1 + 10+l+1) / 2
        assert np.allclose(r[2], [5.0 + l + 1 for l in range(len(r[2]))])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    R, xs, ws, dims, sp, L = _problem()
    full = _update(R, ws, _local_grads(R, sp, ws, dims, L, list(range(len(xs)))), 0.02)
    for l in range(L):
        for a0, a1, ref, w in zip(got[0][l], got[1][l], full[l], (ws[l][0], ws[l][2], ws[l][1], ws[l][3])):
            assert np.array_equal(a0, a1)                                     # replicas agree bit for bit
            dw = max(np.abs(ref - w).max(), 1e-12)
            assert np.abs(a0 - ref).max() < 1e-6 * max(1, np.abs(ref).max()) + 1e-4 * dw


def test_grad_layout_matches_c_packing():
    _setup()
    dp = importlib.import_module("autoencoder-fft_amd.dp")
    dims = [dict(dM=8, dD=3, Nk=5, Nl=5), dict(dM=16, dD=8, Nk=5, Nl=5)]
    lay, n = dp.grad_layout(dims)
    assert n == 2 * 600 + 8 + 3 + 2 * 3200 + 16 + 8                           # SURVEY 8e message sizes: 4.8 KB + 25.7 KB
    assert lay[1]["dck"][0] == 2 * 600 + 11 and lay[1]["dp"] == (lay[1]["dck"][0] + 6400 + 16, 8)
    rng = np.random.default_rng(0)
    per = [tuple(rng.normal(size=s) for s in ((g["dM"], g["dD"], 5, 5), (g["dD"], g["dM"], 5, 5), (g["dM"],), (g["dD"],))) for g in dims]
    back = dp.unpack_grads(dp.pack_grads(per, dims), dims)
    for a, b in zip(per, back):
        for x, y in zip(a, b):
            assert np.array_equal(x.astype(np.float32), y)
