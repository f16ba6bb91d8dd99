"""Round-3 GPU parity tests (through the C ABI): the operator-form MSE in the TRAINED regime against the float64 oracle, layer exports
around a training step (the sequences include/aefft.h documents), the step-form query, a 5-pair tied + multiobjective step against
the oracle for every pair, and config 5 at full size default-vs-literal."""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import np_ref as R
from test_gpu_fft_path import _pair, _step_vs_oracle, host, relerr, weight_step_tol

aefft = importlib.import_module("autoencoder-fft_amd")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = aefft.Context()
    yield c
    c.close()


def _video(rng, B, D, N):
    """frames as the bench makes them: uint8-valued noise plus a smooth component SHARED by every frame (a video's static background):
    the batch mean dominates the low bins -- the regime in which uncentred moments cancel"""
    i = np.arange(N)[:, None] / N
    j = np.arange(N)[None, :] / N
    out = np.empty((B, D, N, N))
    for b in range(B):
        for d in range(D):
            smooth = 64 * (1 + np.sin(2 * np.pi * (i * (d + 1) + 0.3))) * (1 + np.cos(2 * np.pi * j * 2)) / 2
            out[b, d] = np.floor(0.5 * np.floor(rng.uniform(0, 256, (N, N))) + smooth)
    return out


def _near_identity_pair(rng, dD, dM, Nk, eps):
    """encoder / decoder whose composition is the identity up to eps: conv_k divides its input by the number of output maps
    (fft_backproplib.cu:176-177), so c[m][d] = dM * delta(m, d) * delta(centre tap) passes channel d to map d unchanged and
    f[d][m] = dD * delta(d, m) * delta(centre) brings it back.  Noise eps (relative to those weights) on every tap, small biases."""
    assert dM >= dD
    c = np.zeros((dM, dD, Nk, Nk)); f = np.zeros((dD, dM, Nk, Nk))
    for d in range(dD):
        c[d, d, Nk // 2, Nk // 2] = dM
        f[d, d, Nk // 2, Nk // 2] = dD
    c += eps * dM * rng.uniform(-1, 1, c.shape) / (Nk * Nk)
    f += eps * dD * rng.uniform(-1, 1, f.shape) / (Nk * Nk)
    b = eps * rng.uniform(-1, 1, dM); p = eps * rng.uniform(-1, 1, dD)
    q = lambda a: a.astype(np.float32).astype(np.float64)
    return q(c), q(b), q(f), q(p)


@pytest.mark.parametrize("path", ["", "NOCHAIN", "NOOPFORM"])
@pytest.mark.parametrize("maps,s,eps", [([4], 1, 1e-3), ([4, 6], 2, 1e-3), ([4, 6], 2, 1e-4)])
def test_trained_regime_mse_gradients_and_reconstruction_vs_float64_oracle(ctx, flags, path, maps, s, eps):
    """A net near a true optimum (every pair's encoder . decoder = identity + eps): the post-update MSE of a pair is ~eps^2 of the signal
    energy, i.e. five to seven orders of magnitude below it.  mse_d, the packed gradients and the reconstruction of one training step
    against the float64 oracle at the stated tolerances (MSE: 1e-5 * max(1, mse), fft_backproplib.cu:480-498,1178-1192), for the
    operator form with and without the chain launch and for the per-frame form."""
    flags(*path.split(","))
    rng = np.random.default_rng(4 + len(maps))
    D, N, Nk, B = 3, 64, 5, 4
    L = len(maps)
    ws = []
    dD = D
    for dM in maps:
        ws.append(_near_identity_pair(rng, dD, dM, Nk, eps)); dD = dM
    xs = _video(rng, B, D, N)
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    for l, w in enumerate(ws):
        net.set_pair(l, *w)
    assert net.step_form() == {"": "operator_chain", "NOCHAIN": "operator", "NOOPFORM": "per_frame"}[path]
    net_c = [w[0] for w in ws] + [w[2] for w in ws[::-1]]
    net_b = [w[1] for w in ws] + [w[3] for w in ws[::-1]]
    sp = [R.autoenc_fft(xs[i], net_c, net_b, [s] * L + [-s] * L) for i in range(B)]
    recon = ctx.empty(B, D, N, N)
    net.step_grad(ctx.dev(xs), recon)
    for i in range(B):
        assert relerr(host(recon)[i], sp[i][0][-1]) < 1e-4
    gbuf = host(net.grad_buffer()).copy()
    mse = ctx.empty(L)
    # a step far smaller than the distance to the optimum (the default rate moves every tap by +-0.002, i.e. out of the regime under test)
    del0 = 2e-4
    net.step_apply(del0, 0, 0, 1.0, mse)
    off = 0
    for l in range(L):
        c, b, f, p = ws[l]
        dM, dDl = c.shape[:2]
        Xs = [sp[i][2][2 * l + 1] for i in range(B)]
        Os = [sp[i][2][4 * L - 1 - 2 * l] for i in range(B)]
        cf = sp[0][1]
        z = lambda a: np.zeros_like(a)
        r = R.batch_train_iter(Xs, Xs, Os, cf[l], cf[2 * L - 1 - l], c, f, b, p, (z(c), z(f), z(b), z(p)), 0.1 * del0)
        nk = c.size
        # The gradient is linear in the error E = O - X, which float32 forms by cancellation (reference: fft_backproplib.cu:417-424, on
        # float32 spectra): its absolute accuracy is a few ulp of X, whatever the size of E.  Bound: 5e-5 of the gradient itself plus
        # 4e-7 of the gradient the same weights would see with E = -X (nothing reconstructed) -- the float32 floor of this regime.
        gsig = R.batch_grad(Xs, Xs, [np.zeros_like(o) for o in Os], cf[l], cf[2 * L - 1 - l], b, Nk, Nk)
        for seg, ref, sig in zip((gbuf[off:off + nk], gbuf[off + nk:off + 2 * nk], gbuf[off + 2 * nk:off + 2 * nk + dM],
                                  gbuf[off + 2 * nk + dM:off + 2 * nk + dM + dDl]), r["grads"], gsig):
            assert np.abs(seg - ref.ravel()).max() < 5e-5 * np.abs(ref).max() + 4e-7 * np.abs(sig).max(), (l, np.abs(seg - ref.ravel()).max(), np.abs(ref).max(), np.abs(sig).max())
        off += 2 * nk + dM + dDl
        got = float(host(mse)[l])
        # the signal energy in the MSE's own normalisation: what the MSE would be if the pair output nothing
        x_energy = float(np.mean([R.mse_fft(X, np.zeros_like(X), dM, dDl, X.shape[-2], (X.shape[-1] - 1) * 2) for X in Xs]))
        assert r["mse"] < 1e-4 * x_energy, "the test must sit in the trained regime"
        assert abs(got - r["mse"]) < 1e-5 * max(1.0, r["mse"]), (l, got, r["mse"])
        assert abs(got - r["mse"]) < 2e-2 * r["mse"] + 1e-9 * x_energy, (l, got, r["mse"], x_energy)      # ... and relative to the MSE itself
    net.close()


@pytest.mark.parametrize("path", ["NOOPFORM,GTAPS", "NOCHAIN,NOFUSEUPD,GTAPS"])
def test_gprime_from_taps_with_many_channels(ctx, flags, path):
    """The post-update MSE through G' = F'.C'/(dM dD) formed from the taps (the route HBM-sized spectra take, forced by GTAPS) on a
    middle pair of 32 -> 40 maps, per-frame and operator form, against the oracle (round 2's separate tap-product kernel was wrong
    beyond 28 input channels and only ever tested below that)."""
    flags(*path.split(","))
    _step_vs_oracle(ctx, np.random.default_rng(5), 2, 3, 64, 64, [32, 40, 8], 5, 2)


@pytest.mark.parametrize("path", ["NOOPFORM,GTAPS", "NOCHAIN,NOFUSEUPD,GTAPS"])
def test_gprime_taps_formed_once_for_row_chunked_planes(ctx, flags, path):
    """... on grids of more than 64 rows, where a plane of G' is transformed by several row-chunk workgroups and its (2Nk-1)^2 taps come from
    a launch in front (gtaps_group_kernel) instead of being formed again by each of them: two pairs on a 128 x 128 grid (no pooling)."""
    flags(*path.split(","))
    _step_vs_oracle(ctx, np.random.default_rng(6), 2, 3, 128, 128, [6, 10], 5, 1)


def _small_net(ctx, rng, D=3, N=64, maps=(4, 6, 5), Nk=5, s=2, B=3):
    net = aefft.Net(ctx, D, N, N, list(maps), Nk, s, batch=B)
    dD = D
    for l, dM in enumerate(maps):
        _, cw, fw, bw, pw = _pair(rng, dD, dM, 8, Nk, 1)
        net.set_pair(l, cw, bw, fw, pw); dD = dM
    return net


def test_layer_exports_around_a_training_step(ctx, flags):
    """include/aefft.h, aefft_net_get_layer: after step_grad the layers are those of that step's forward (weights before the update).
    (i) in the chain form they survive step_apply AND the caller overwriting its frame buffer; (ii) a get_layer call between
    step_grad and step_apply changes nothing of the step; (iii) every non-hidden layer equals the per-frame form's."""
    L = 3
    x_np = np.floor(np.random.default_rng(11).uniform(0, 256, (3, 3, 64, 64)))
    nonhidden = [l for l in range(1, 4 * L + 1) if not (l <= 2 * L and l % 2 == 0)]
    ref = {}
    for path in ("NOOPFORM", ""):
        flags(*path.split(","))
        net = _small_net(ctx, np.random.default_rng(10))
        x = ctx.dev(x_np)
        recon, mse = ctx.empty(*x.shape), ctx.empty(L)
        net.step_grad(x, recon)
        between = {l: host(net.get_layer(l)).copy() for l in nonhidden}            # between the two halves
        net.step_apply(0.2, 0, 0, 1.0, mse)
        weights = [net.get_pair(l) for l in range(L)]
        if path == "":
            assert net.step_form() == "operator_chain"
            x.fill_(7.0)                                                           # the caller reuses its frame buffer
            after = {l: host(net.get_layer(l)).copy() for l in nonhidden}
            for l in nonhidden:
                assert np.array_equal(after[l], between[l]), l                     # same operators, same resident spectra
        ref[path] = (between, weights, host(mse).copy())
        net.close()
    for l in nonhidden:
        assert relerr(ref[""][0][l], ref["NOOPFORM"][0][l]) < 5e-5, l
    # (ii): the step itself against a run without the export in between
    flags()
    net = _small_net(ctx, np.random.default_rng(10))
    mse = ctx.empty(L)
    net.step_grad(ctx.dev(x_np), None); net.step_apply(0.2, 0, 0, 1.0, mse)
    for wa, wb in zip(ref[""][1], [net.get_pair(l) for l in range(L)]):
        for a, b in zip(wa, wb):
            assert np.array_equal(a, b)
    assert np.allclose(host(mse), ref[""][2], rtol=1e-6)
    net.close()


def test_step_form_query(ctx, flags):
    """aefft_net_step_form names the form a net's training step runs in (no silent fallback)"""
    mk = lambda D, maps, Nk, N=32, s=2: aefft.Net(ctx, D, N, N, maps, Nk, s, batch=2)
    for args, want in (((3, [4, 6], 5), "operator_chain"), ((1, [4], 3), "operator_chain"), ((4, [4, 6], 5), "per_frame"),
                       ((3, [4], 7), "per_frame")):
        net = mk(*args)
        assert net.step_form() == want, (args, net.step_form())
        net.close()
    net = mk(3, [4, 6], 5)
    for f, want in (("NOOPFORM", "per_frame"), ("NOQPATH", "per_frame"), ("NOCHAIN", "operator"), ("NOMFMA", "operator"), ("", "operator_chain")):
        flags(*f.split(","))
        assert net.step_form() == want, f
    net.close()


@pytest.mark.parametrize("path", ["", "NOOPFORM"])
def test_five_pair_tied_multiobjective_step_vs_oracle_every_pair(ctx, flags, path):
    """config 5's structure at reduced planes and maps (256x256, 5 pairs 8/16/24/32/48 maps, 5x5, pool 2, symmetric weights + multiobjective): ONE
    step against np_ref.batch_grad / gradient_diff / backprop_sym for EVERY pair (fft_backproplib.cu:657-753; the tied rule is
    build-defined, SURVEY B-14, mirroring backproplib.cu:533,619-622).  The multi-pair tied / multiobjective step runs the separate
    update launch (update_group_kernel / per-pair gradient_diff), which no other test compares with anything."""
    flags(*path.split(","))
    rng = np.random.default_rng(55)
    # (map counts halved against config 5 from the third pair on: the oracle's gradient_diff is O((dM dD)^2 Nk Nl) in numpy)
    D, N, maps, Nk, s, B = 3, 256, [8, 16, 24, 32, 48], 5, 2, 2
    L = len(maps)
    ws = []
    dD = D
    for dM in maps:
        c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32).astype(np.float64)
        b = rng.uniform(-1, 1, dM).astype(np.float32).astype(np.float64)
        p = rng.uniform(-1, 1, dD).astype(np.float32).astype(np.float64)
        ws.append((c, b, np.transpose(c, (1, 0, 2, 3)).copy(), p)); dD = dM
    xs = np.floor(rng.uniform(0, 256, (B, D, N, N)))
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    for l, w in enumerate(ws):
        net.set_pair(l, *w)
    net.step_grad(ctx.dev(xs), None)
    mse = ctx.empty(L)
    net.step_apply(0.2, 1, 1, 1.0, mse)
    assert np.isfinite(host(mse)).all()
    net_c = [w[0] for w in ws] + [w[2] for w in ws[::-1]]
    net_b = [w[1] for w in ws] + [w[3] for w in ws[::-1]]
    sp = [R.autoenc_fft(x, net_c, net_b, [s] * L + [-s] * L) for x in xs]
    z = lambda a: np.zeros_like(a)
    for l in range(L):
        c, b, f, p = ws[l]
        Xs = [q[2][2 * l + 1] for q in sp]; Os = [q[2][4 * L - 1 - 2 * l] for q in sp]
        dck, dfk, db, dp = R.batch_grad(Xs, Xs, Os, sp[0][1][l], sp[0][1][2 * L - 1 - l], b, Nk, Nk)
        extra = R.gradient_diff(c, f, b, p)
        rc, rf, rb, rp = R.backprop_sym(c, f, b, p, dck, dfk, db, dp, z(c), z(f), z(b), z(p), 0.02, *extra)[:4]
        c2, b2, f2, p2 = net.get_pair(l)
        assert np.array_equal(f2, np.transpose(c2, (1, 0, 2, 3))), l
        g_used = 0.5 * (dck + np.transpose(dfk, (1, 0, 2, 3))) - 10.0 * 0.5 * (extra[0] + np.transpose(extra[1], (1, 0, 2, 3)))
        assert (np.abs(c2 - rc) < weight_step_tol(g_used)).all(), (l, np.abs(c2 - rc).max())
        assert (np.abs(b2 - rb) < weight_step_tol(0.5 * db - 10.0 * extra[2], grel=2e-4)).all(), l
        assert (np.abs(p2 - rp) < weight_step_tol(0.5 * dp - 10.0 * extra[3], grel=2e-4)).all(), l
        assert np.abs(c2 - c).max() > 1e-4
    net.close()


LITERAL = ["NOOPFORM", "NOLAZY", "NOCOMPACT", "NOQPATH", "NOFUSEMSE", "NOGROUP", "NOMFMA", "NOGFWD", "NOOVERLAP", "NOFUSECROP"]


def test_config5_full_size_default_equals_literal(ctx, flags):
    """BASELINE configs[4] at full size (1024x1024, 5 pairs 8/16/32/64/128 maps, 5x5, pool 2, symmetric weights + multiobjective), B = 4
    frames: one training step of the default path against the same library with every re-association switched off (per-frame conv /
    S / dc,df / C2R / update / R2C / conv, conv / MSE on the scalar-FMA kernels) -- reconstruction, packed gradients per pair, the tied
    weights after the update, the post-update MSE per pair."""
    rng = np.random.default_rng(1055)
    D, N, maps, Nk, s, B = 3, 1024, [8, 16, 32, 64, 128], 5, 2, 4
    L = len(maps)
    ws = []
    dD = D
    for dM in maps:
        c = rng.uniform(-1, 1, (dM, dD, Nk, Nk))
        ws.append((c, rng.uniform(-1, 1, dM), np.transpose(c, (1, 0, 2, 3)).copy(), rng.uniform(-1, 1, dD))); dD = dM
    frames = ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N))))
    res = []
    for literal in (True, False):
        flags(*(LITERAL if literal else []))
        net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
        for l, w in enumerate(ws):
            net.set_pair(l, *w)
        if not literal:
            assert net.step_form() == "operator_chain"
        recon, mse = ctx.empty(B, D, N, N), ctx.empty(L)
        net.step_grad(frames, recon)
        g = host(net.grad_buffer()).copy()
        net.step_apply(0.2, 1, 1, 1.0, mse)
        res.append((g, [net.get_pair(l) for l in range(L)], host(recon).copy(), host(mse).copy()))
        net.close()
    (g_lit, w_lit, r_lit, m_lit), (g_opt, w_opt, r_opt, m_opt) = res
    assert relerr(r_opt, r_lit) < 2e-5
    off = 0
    for l, (c, b, f, p) in enumerate(ws):
        n = 2 * c.size + b.size + p.size
        assert relerr(g_opt[off:off + n], g_lit[off:off + n]) < 5e-5, l
        off += n
    for l, (a, b_) in enumerate(zip(w_lit, w_opt)):
        c0 = ws[l][0]
        assert np.abs(a[0] - c0).max() > 1e-4                                        # the update happened
        # clipped entries move by exactly +-0.002 on both sides; the few whose multiobjective gradient lies at the knee |g| = 10 (or near 0)
        # inherit the gradients' 5e-5: all but a thousandth of the entries agree to rounding, none is off by more than two steps
        dcw = np.abs(a[0] - b_[0])
        assert np.mean(dcw < 1e-6 + 2e-3 * np.abs(a[0] - c0).max()) > 0.999 and dcw.max() < 4.1e-3, (l, dcw.max())
        assert np.array_equal(b_[2], np.transpose(b_[0], (1, 0, 2, 3))), l
        for x, y in zip(a[1:], b_[1:]):
            assert np.abs(x - y).max() < 4.1e-3, l
    assert np.allclose(m_lit, m_opt, rtol=2e-4), (m_lit, m_opt)


def test_c_level_step_with_rccl_on_the_library_stream():
    """INTEGRATION.md section 3 from plain C++ (examples/rccl_step.cpp, no torch): aefft_net_step_grad -> ncclAllReduce of
    aefft_net_grad_buffer on aefft_stream(ctx) -> aefft_net_step_apply, world size 1 (the only size a one-GPU box allows): the
    collective and both halves ordered on the library's stream, the packed buffer's MSE tail = the previous step's MSE, weights equal
    to a net trained without the collective."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "autoencoder-fft_amd", "aefft_rccl_step")
    assert os.path.exists(exe), "build it: make -C autoencoder-fft_amd/csrc rccl_step (__graft_entry__.build does)"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), RANK="0", WORLD_SIZE="1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "rccl_step ok" in out.stdout, (out.stdout, out.stderr[-2000:])


def test_step_apply_saves_the_reduced_mse_tail_behind_the_packed_buffer(ctx):
    """SURVEY 8e / include/aefft.h: the packed buffer's tail carries the previous step's post-update MSEs through the caller's all-reduce;
    aefft_net_step_apply keeps what it finds there, times grad_scale, in the L floats behind the buffer before its own values take the
    tail -- so that no copy kernel has to be enqueued between the collective and the update half."""
    rng = np.random.default_rng(77)
    D, N, maps, Nk, s, B = 3, 32, [4, 6], 5, 2, 2
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    dD = D
    for l, dM in enumerate(maps):
        net.set_pair(l, rng.uniform(-1, 1, (dM, dD, Nk, Nk)), rng.uniform(-1, 1, dM), rng.uniform(-1, 1, (dD, dM, Nk, Nk)), rng.uniform(-1, 1, dD)); dD = dM
    x = ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N))))
    mse = ctx.empty(len(maps))
    L = len(maps)
    net.step_grad(x, None); net.step_apply(0.02, 0, 0, 1.0, mse)
    m1 = host(mse).copy()
    assert np.array_equal(host(net.mse_prev_global()), np.zeros(L, np.float32))          # nothing before the first step
    assert np.array_equal(host(net.grad_buffer())[-L:], m1)
    net.step_grad(x, None)
    net.grad_buffer()[-L:] *= 2.0                                                          # a two-rank all-reduce (SUM) of equal tails
    net.step_apply(0.02, 0, 0, 0.5, mse)
    m2 = host(mse).copy()
    assert np.allclose(host(net.mse_prev_global()), m1, rtol=1e-6)                          # global mean of step 1, saved by step 2's update half
    assert np.array_equal(host(net.grad_buffer())[-L:], m2)
    net.close()


@pytest.mark.parametrize("path", ["", "NOOPFORM", "NOLAZYMSE"])
def test_step_apply_without_mse_output_defers_the_sums_to_the_next_gradient_launch(ctx, flags, path):
    """aefft_net_step_apply(mse_d = NULL) leaves the slot sums of the post-update MSE to one extra workgroup of the next step's gradient
    launch (or to aefft_net_last_mse): weights, the packed buffer's MSE tail as the all-reduce sees it, the saved global MSE and the MSEs
    themselves are those of the run that asks for the MSE every step -- also when a forward / layer export / burst sits in between."""
    rng = np.random.default_rng(55)
    D, N, maps, Nk, s, B = 3, 64, [4, 6, 5], 5, 2, 2
    L = len(maps)
    ws, dD = [], D
    for dM in maps:
        ws.append((rng.uniform(-1, 1, (dM, dD, Nk, Nk)), rng.uniform(-1, 1, dM), rng.uniform(-1, 1, (dD, dM, Nk, Nk)), rng.uniform(-1, 1, dD))); dD = dM
    xs = [ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N)))) for _ in range(4)]
    res = []
    for ask in (True, False):
        flags(*path.split(","))
        net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
        for l, w in enumerate(ws):
            net.set_pair(l, *w)
        mse = ctx.empty(L)
        tails, prevs, mses = [], [], []
        for i, x in enumerate(xs):
            net.step_grad(x, None)
            tails.append(host(net.grad_buffer())[-L:].copy())          # what an all-reduce would carry: the previous step's MSEs
            prevs.append(host(net.mse_prev_global()).copy())            # (saved when those sums were formed: the step before that)
            net.step_apply(0.02, 0, 0, 1.0, mse if ask else None)
            if i == 1:
                net.get_layer(2)                                        # something else between two steps
            if ask or i == len(xs) - 1:
                if not ask:
                    net.last_mse(mse)
                mses.append(host(mse).copy())
        res.append((tails, prevs, mses, [net.get_pair(l) for l in range(L)]))
        net.close()
    (t0, p0, m0, w0), (t1, p1, m1, w1) = res
    for a, b in zip(t0, t1):
        assert np.array_equal(a, b)
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)
    assert np.array_equal(m0[-1], m1[-1]) and np.all(m0[-1] > 0)
    for a, b in zip(w0, w1):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
