"""Device data-parallel path: two processes share the one GPU of the test box, each trains on ITS half of the
frames through the C ABI; the packed gradient buffers are all-reduced (gloo on CUDA tensors stands in for RCCL,
which needs one GPU per rank) and both ranks must end up with the weights of a single process that saw all frames."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    rng = np.random.default_rng(7)
    D, N, maps, Nk, s, B = 3, 32, [4, 6], 5, 2, 4
    xs = np.floor(rng.uniform(0, 256, (B, D, N, N))).astype(np.float32)
    ws, dD = [], D
    for dM in maps:
        ws.append((rng.uniform(-1, 1, (dM, dD, Nk, Nk)), rng.uniform(-1, 1, dM), rng.uniform(-1, 1, (dD, dM, Nk, Nk)), rng.uniform(-1, 1, dD)))
        dD = dM
    return D, N, maps, Nk, s, xs, ws


def _train(aefft, dp, frames, B, steps=3, pipelined=False):
    D, N, maps, Nk, s, _, ws = _problem()
    ctx = aefft.Context(0)
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    for l, w in enumerate(ws):
        net.set_pair(l, *w)
    net.set_input_ready(pipelined)       # as bench.py does at N > 1: input R2C and reconstruction C2R move into the all-reduce gap
    step = dp.DataParallelStep(net)
    fr = ctx.dev(frames)
    mse = ctx.empty(len(maps))
    recon = ctx.empty(*fr.shape) if pipelined else None
    for _ in range(steps):
        step(fr, recon, 0.2, 0, 0, mse)
    ctx.sync()
    ok = step.replicas_agree()
    out = [net.get_pair(l) for l in range(len(maps))]
    net.close(); ctx.close()
    return out, ok


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    aefft = importlib.import_module("autoencoder-fft_amd"); dp = importlib.import_module("autoencoder-fft_amd.dp")
    xs = _problem()[5]
    per = len(xs) // world
    out, ok = _train(aefft, dp, xs[rank * per:(rank + 1) * per], per, pipelined=True)
    q.put((rank, out, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_device_equal_single_process():
    world, port = 2, 29600 + (os.getpid() % 2000)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, out, ok = q.get(timeout=300)
        got[r] = out
        assert ok
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    aefft = importlib.import_module("autoencoder-fft_amd"); dp = importlib.import_module("autoencoder-fft_amd.dp")
    xs, ws = _problem()[5], _problem()[6]
    ref, _ = _train(aefft, dp, xs, len(xs))
    for l in range(len(ref)):
        for a0, a1, r, w in zip(got[0][l], got[1][l], ref[l], ws[l]):
            assert np.array_equal(a0, a1)                                       # replicas bit-identical
            dw = max(np.abs(r - w.astype(np.float32)).max(), 1e-9)
            assert np.abs(a0 - r).max() < 1e-6 + 2e-3 * dw
