"""CPU-path checkers: oracle/cpu_ref.c (port) must be bit-identical to the reference's own
Conv/backprop/Pool/Portion compiled from /root/reference (oracle/_ref), and both must
reproduce the committed golden vectors (tests/golden/cpu_path.npz, made by
tests/golden/make_golden.py from the compiled reference)."""
import os

import numpy as np
import pytest

import cpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "cpu_path.npz")


def _case(seed, dD, dM, N, Nk):
    rng = np.random.default_rng(seed)
    x = np.floor(rng.uniform(0, 256, (dD, N, N))).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32)
    f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    b = rng.uniform(-1, 1, dM).astype(np.float32)
    p = rng.uniform(-1, 1, dD).astype(np.float32)
    return x, c, b, f, p


@pytest.mark.parametrize("shape", [(1, 4, 16, 3), (3, 4, 12, 5), (2, 3, 10, 3), (2, 2, 9, 7)])
def test_port_bit_identical_to_compiled_reference(shape):
    ref = cpu.reference()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    port = cpu.port()
    x, c, b, f, p = _case(7, *shape)
    h1, h2 = port.conv(x, c, b), ref.conv(x, c, b)
    assert np.array_equal(h1, h2)
    o1, o2 = port.conv(h1, f, p), ref.conv(h2, f, p)
    assert np.array_equal(o1, o2)
    for a, r in zip(port.backprop(x, o1, h1, c, b, f, p, 0.2), ref.backprop(x, o2, h2, c, b, f, p, 0.2)):
        assert np.array_equal(a, r)
    dD, N = x.shape[0], x.shape[1]
    for s in (1, 2, 3):
        assert np.array_equal(port.pool(x, (dD, -(-N // s), -(-N // s)), s), ref.pool(x, (dD, -(-N // s), -(-N // s)), s))
    assert np.array_equal(port.pool(x, (dD, 2 * N, 2 * N), -2), ref.pool(x, (dD, 2 * N, 2 * N), -2))
    for q in (1, 2):
        assert np.array_equal(port.portion(x, q), ref.portion(x, q))


def test_pool_truncates_and_clamps():
    """netlib.cpp:127-136: `int smax=0` -> values truncated to int, negatives clamp to 0, even at scale 1."""
    port = cpu.port()
    x = np.array([[[1.9, -3.0], [2.5, 7.99]]], np.float32)
    assert np.array_equal(port.pool(x, (1, 2, 2), 1), np.array([[[1, 0], [2, 7]]], np.float32))
    assert np.array_equal(port.pool(x, (1, 1, 1), 2), np.array([[[7]]], np.float32))


def test_conv_excludes_row_col_zero():
    """netlib.cpp:344: boundary test '>0' -- input row 0 / col 0 never contributes."""
    port = cpu.port()
    x = np.zeros((1, 6, 6), np.float32)
    x[0, 0, :] = 5; x[0, :, 0] = 7
    c = np.ones((1, 1, 3, 3), np.float32)
    assert np.all(port.conv(x, c, np.zeros(1, np.float32)) == 0)


@pytest.mark.parametrize("lib", ["port", "reference"])
def test_golden_cpu_path(lib):
    L = cpu.port() if lib == "port" else cpu.reference()
    if L is None:
        pytest.skip("oracle/_ref not built")
    g = np.load(GOLD)
    for tag in ("cfg1", "k5"):
        x, c, b, f, p = (g[f"{tag}_{n}"] for n in ("x", "c", "b", "f", "p"))
        pin = L.pool(x, x.shape, 1)
        h = L.conv(pin, c, b)
        o = L.conv(h, f, p)
        assert np.array_equal(h, g[f"{tag}_h"]) and np.array_equal(o, g[f"{tag}_o"])
        c2, b2, f2, p2 = L.backprop(pin, o, h, c, b, f, p, 0.2)
        for a, n in ((c2, "c2"), (b2, "b2"), (f2, "f2"), (p2, "p2")):
            assert np.array_equal(a, g[f"{tag}_{n}"]), n
