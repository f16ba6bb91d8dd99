"""The C++ operator-API boundary (include/netlib.h, backproplib.h, fft_backproplib.h): exported mangled
symbols, the host-side functions against the compiled reference (CPU), and the vector entry points
against the oracle (GPU)."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

import cpu
import np_ref as R
import np_spatial as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
aefft = importlib.import_module("autoencoder-fft_amd")
FP = C.POINTER(C.c_float)

# the symbols autoencoder.cpp imports (libstdc++ mangling)
WANTED = """
_Z11autoenc_fftRSt6vectorIS_IS_IS_IfSaIfEESaIS1_EESaIS3_EESaIS5_EERS_IS7_SaIS7_EERS3_SC_RS_IiSaIiEEi
_Z12backprop_fftRSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EES6_S6_RS1_RS_IS5_SaIS5_EES7_SA_S7_S7_ifi
_Z12backprop_gpuRSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EES6_S6_RS_IS5_SaIS5_EERS1_S9_SA_S9_SA_S9_SA_S9_SA_S9_SA_ffi
_Z13ImageToSpin_CRN2cv3MatERSt6vectorIS2_IS2_IfSaIfEESaIS4_EESaIS6_EE
_Z13SaveLoad_convRSt6vectorIS_IS_IS_IfSaIfEESaIS1_EESaIS3_EESaIS5_EERS1_iiii
_Z13SpinToImage_CRN2cv3MatERSt6vectorIS2_IS2_IfSaIfEESaIS4_EESaIS6_EE
_Z13SpinToImage_KRN2cv3MatERSt6vectorIS2_IfSaIfEESaIS4_EE
_Z13SpinToImage_VRN2cv3MatERSt6vectorIS2_IfSaIfEESaIS4_EE
_Z15backprop_gpu_ccRSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EES6_S6_RS_IS5_SaIS5_EERS1_S9_SA_S9_SA_S9_SA_S9_SA_S9_SA_ffi
_Z3actf
_Z4PoolRSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EES6_i
_Z4act1f
_Z7PortionRSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EES6_S6_S6_S6_S6_i
_Z8Conv_gpuRSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EES6_RS_IS5_SaIS5_EERS1_
_Z8backpropRSt6vectorIS_IS_IfSaIfEESaIS1_EESaIS3_EES6_S6_RS_IS5_SaIS5_EERS1_S9_SA_f
_Z9Init_convRSt6vectorIS_IS_IS_IfSaIfEESaIS1_EESaIS3_EESaIS5_EERS1_iiiif
_Z9LoadParamRiS_S_S_Rf
""".split()


def _p(a):
    return a.ctypes.data_as(FP)


@pytest.fixture(scope="module")
def wrap():
    if not os.path.exists(aefft.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    out = os.path.join(ROOT, "tests", "_shim_wrap.so")
    src = os.path.join(ROOT, "tests", "shim_wrap.cpp")
    libdir = os.path.dirname(aefft.LIB_PATH)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(aefft.LIB_PATH)):
        subprocess.run(["g++", "-O2", "-std=c++11", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
                        "-L" + libdir, "-laefft", "-Wl,-rpath," + libdir], check=True)
    L = C.CDLL(out)
    for n in ("w_conv", "w_backprop_cpu", "w_pool", "w_portion", "w_backprop_gpu", "w_fft_pair", "w_init_conv", "w_saveload_conv", "w_load_param"):
        getattr(L, n).restype = None
    return L


def test_reference_symbols_exported():
    out = subprocess.run(["nm", "-D", "--defined-only", aefft.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines())
    assert not [s for s in WANTED if s not in exported]


def test_headers_declare_the_reference_manglings(tmp_path):
    """A TU that only sees OUR headers must reference exactly the reference's symbols."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "netlib.h"\n#include "backproplib.h"\n#include "fft_backproplib.h"\n'
                   "void* tab[] = {(void*)&autoenc_fft,(void*)&backprop_fft,(void*)&backprop_gpu,(void*)&ImageToSpin_C,(void*)&SaveLoad_conv,"
                   "(void*)&SpinToImage_C,(void*)&SpinToImage_K,(void*)&SpinToImage_V,(void*)&backprop_gpu_cc,(void*)&act,(void*)&Pool,(void*)&act1,"
                   "(void*)&Portion,(void*)&Conv_gpu,(void*)&backprop,(void*)&Init_conv,(void*)&LoadParam};\n")
    obj = tmp_path / "t.o"
    subprocess.run(["g++", "-std=c++11", "-I" + os.path.join(ROOT, "include"), "-c", str(src), "-o", str(obj)], check=True)
    und = subprocess.run(["nm", "-u", str(obj)], capture_output=True, text=True, check=True).stdout
    assert sorted(l.split()[-1] for l in und.splitlines() if "_Z" in l) == sorted(WANTED)


def test_host_functions_bit_identical_to_compiled_reference(wrap):
    ref = cpu.reference()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    for dD, dM, N, Nk in ((1, 4, 16, 3), (3, 2, 12, 5)):
        x = np.floor(rng.uniform(0, 256, (dD, N, N))).astype(np.float32)
        c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
        b = rng.uniform(-1, 1, dM).astype(np.float32); p = rng.uniform(-1, 1, dD).astype(np.float32)
        h = np.zeros((dM, N, N), np.float32)
        wrap.w_conv(_p(x), _p(h), _p(c), _p(b), dD, dM, N, N, Nk, Nk, 0)
        assert np.array_equal(h, ref.conv(x, c, b))
        o = np.zeros((dD, N, N), np.float32)
        wrap.w_conv(_p(h), _p(o), _p(f), _p(p), dM, dD, N, N, Nk, Nk, 0)
        c2, b2, f2, p2 = c.copy(), b.copy(), f.copy(), p.copy()
        wrap.w_backprop_cpu(_p(x), _p(o), _p(h), _p(c2), _p(b2), _p(f2), _p(p2), C.c_float(0.2), dD, dM, N, N, Nk, Nk)
        for a, r in zip((c2, b2, f2, p2), ref.backprop(x, o, h, c, b, f, p, 0.2)):
            assert np.array_equal(a, r)
        pd = np.zeros((dD, N // 2, N // 2), np.float32)
        wrap.w_pool(_p(x), _p(pd), dD, N, N, N // 2, N // 2, 2)
        assert np.array_equal(pd, ref.pool(x, (dD, N // 2, N // 2), 2))
        pu = np.zeros((dD, 2 * N, 2 * N), np.float32)
        wrap.w_pool(_p(x), _p(pu), dD, N, N, 2 * N, 2 * N, -2)
        assert np.array_equal(pu, ref.pool(x, (dD, 2 * N, 2 * N), -2))
        ps = np.zeros((dD, N // 2, N // 2), np.float32)
        wrap.w_portion(_p(x), _p(ps), dD, N, N, 2)
        assert np.array_equal(ps, ref.portion(x, 2))


@pytest.mark.parametrize("tag,dD,dM,N,Nk", [("cfg1", 1, 4, 128, 3), ("k5", 2, 3, 20, 5)])
def test_config1_golden_vectors_through_the_product_host_functions(wrap, tag, dD, dM, N, Nk):
    """BASELINE configs[0] (128x128 gray, 4 maps, 3x3, spatial mode, CPU path) through the PRODUCT's host functions -- the
    `Pool`, `Conv`, `Portion`, `backprop` that libaefft.so exports under the reference's manglings (shims.cpp; reference:
    netlib.cpp:114-164, 292-315, 318-451) -- in the application's call order Pool(1) -> Conv -> Conv -> Pool(-1) -> Portion(q=1)
    -> backprop(del=0.2) (autoencoder.cpp:135-150,165-182), against tests/golden/cpu_path.npz, whose outputs come from the
    reference's own compiled code (tests/golden/make_golden.py).  Bit-exact: float32 host arithmetic in the reference's loop order."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "cpu_path.npz"))
    x, c, b, f, p = (np.ascontiguousarray(g[f"{tag}_{k}"]) for k in ("x", "c", "b", "f", "p"))
    assert x.shape == (dD, N, N) and c.shape == (dM, dD, Nk, Nk)
    pin = np.zeros_like(x)
    wrap.w_pool(_p(x), _p(pin), dD, N, N, N, N, 1)                       # Pool(+1): the int-truncating maximum (SURVEY B-16)
    h = np.zeros((dM, N, N), np.float32)
    wrap.w_conv(_p(pin), _p(h), _p(c), _p(b), dD, dM, N, N, Nk, Nk, 0)
    assert np.array_equal(h, g[f"{tag}_h"])
    o = np.zeros((dD, N, N), np.float32)
    wrap.w_conv(_p(h), _p(o), _p(f), _p(p), dM, dD, N, N, Nk, Nk, 0)
    assert np.array_equal(o, g[f"{tag}_o"])
    up = np.zeros_like(o)
    wrap.w_pool(_p(o), _p(up), dD, N, N, N, N, -1)                       # Pool(-1): nearest-neighbour up-sampling by 1
    assert np.array_equal(up, o)
    ps = np.zeros_like(pin)
    wrap.w_portion(_p(pin), _p(ps), dD, N, N, 1)                         # Portion(q = 1): the whole plane
    assert np.array_equal(ps, pin)
    c2, b2, f2, p2 = c.copy(), b.copy(), f.copy(), p.copy()
    wrap.w_backprop_cpu(_p(ps), _p(up), _p(h), _p(c2), _p(b2), _p(f2), _p(p2), C.c_float(0.2), dD, dM, N, N, Nk, Nk)
    for a, k in ((c2, "c2"), (b2, "b2"), (f2, "f2"), (p2, "p2")):
        assert np.array_equal(a, g[f"{tag}_{k}"]), k
    assert np.abs(c2 - c).max() > 0


def _ref_lib():
    ref = cpu.reference()
    if ref is None or not hasattr(ref.lib, "ref_init_conv"):
        pytest.skip("oracle/_ref not built (or built from the older recipe)")
    for n in ("ref_init_conv", "ref_saveload_conv", "ref_load_param"):
        getattr(ref.lib, n).restype = None
    return ref.lib


def test_init_conv_draws_the_reference_rand_stream(wrap):
    """`Init_conv` (netlib.cpp:166-197): weights U(-max, max) drawn from the C library's rand() in the order m, d, k, l, then the
    bias of map m.  Same srand seed -> the product's export and the reference's compiled function return the same bits."""
    R_ = _ref_lib()
    libc = C.CDLL(None)
    for seed, (mS, dD, kS, lS, mx) in enumerate(((4, 3, 5, 5, 3.0), (10, 1, 3, 3, 1.0), (2, 7, 3, 5, 0.25))):
        cr = np.zeros((mS, dD, kS, lS), np.float32); br = np.zeros(mS, np.float32)
        cp, bp = np.zeros_like(cr), np.zeros_like(br)
        libc.srand(1234 + seed); R_.ref_init_conv(_p(cr), _p(br), mS, dD, kS, lS, C.c_float(mx))
        libc.srand(1234 + seed); wrap.w_init_conv(_p(cp), _p(bp), mS, dD, kS, lS, C.c_float(mx))
        assert np.array_equal(cr, cp) and np.array_equal(br, bp)
        assert np.abs(cr).max() <= mx and np.abs(cr).max() > 0.5 * mx


def test_weight_files_and_layer_parameters_against_the_compiled_reference(wrap, tmp_path, capfd):
    """`SaveLoad_conv` (netlib.cpp:220-272) and `LoadParam` (:274-289): file NAMES (./weights/C_weights_<L>_<in|out>_D=_M=_Lk=_Ll=_S=.conv)
    and BYTES written by the product == the reference's; each side loads the file the other wrote; New_Layer_Param.txt parsed alike."""
    R_ = _ref_lib()
    rng = np.random.default_rng(31)
    cwd = os.getcwd()
    dirs = {k: tmp_path / k for k in ("ref", "prod")}
    try:
        for dM, dD, Nk, Nl, scale, L, io in ((4, 3, 5, 5, 2, 0, 0), (3, 4, 5, 5, -2, 0, 1), (10, 1, 3, 7, 1, 3, 0)):
            c = rng.uniform(-3, 3, (dM, dD, Nk, Nl)).astype(np.float32); b = rng.uniform(-3, 3, dM).astype(np.float32)
            for k, fn in (("ref", R_.ref_saveload_conv), ("prod", wrap.w_saveload_conv)):
                (dirs[k] / "weights").mkdir(parents=True, exist_ok=True)
                os.chdir(dirs[k])
                fn(_p(c.copy()), _p(b.copy()), dM, dD, Nk, Nl, scale, L, io, 1)
            files = {k: sorted(os.listdir(dirs[k] / "weights")) for k in dirs}
            assert files["ref"] == files["prod"]
            name = f"C_weights_{L}_{'in' if io == 0 else 'out'}_D={dD}_M={dM}_Lk={(Nk - 1) // 2 - 1}_Ll={(Nl - 1) // 2 - 1}_S={scale}.conv"
            assert name in files["ref"]
            raw = {k: (dirs[k] / "weights" / name).read_bytes() for k in dirs}
            assert raw["ref"] == raw["prod"] and len(raw["ref"]) == 4 * (c.size + dM)
            assert np.array_equal(np.frombuffer(raw["ref"], np.float32), np.concatenate([c.ravel(), b]))
            # cross-load: the product reads the reference's file, the reference reads the product's
            for k, fn in (("ref", wrap.w_saveload_conv), ("prod", R_.ref_saveload_conv)):
                os.chdir(dirs[k])
                c2 = np.zeros_like(c); b2 = np.zeros_like(b)
                fn(_p(c2), _p(b2), dM, dD, Nk, Nl, scale, L, io, 0)
                assert np.array_equal(c2, c) and np.array_equal(b2, b)
        # LoadParam: the shipped file's format (name value per line; values: maps, Lk, Ll, pooling scale, rmax)
        os.chdir(dirs["ref"])
        (dirs["ref"] / "New_Layer_Param.txt").write_text("Layer_depth 10\nKernel_L_x 1\nKernel_L_y 0\nPooling_scale 2\nMax_Rand_Init 3.5\n")
        got = {}
        for k, fn in (("ref", R_.ref_load_param), ("prod", wrap.w_load_param)):
            iv = [C.c_int(-1) for _ in range(4)]; fv = C.c_float(-1)
            fn(*[C.byref(v) for v in iv], C.byref(fv))
            got[k] = [v.value for v in iv] + [fv.value]
        assert got["ref"] == got["prod"] == [10, 1, 0, 2, 3.5]
    finally:
        os.chdir(cwd)
    capfd.readouterr()      # (both sides print "path ..." lines, netlib.cpp:235)


@pytest.mark.gpu
def test_vector_entry_points_fft_mode(wrap):
    """autoenc_fft (fft_l=1, cache miss then hit) + backprop_fft (100 iterations) through nested vectors."""
    rng = np.random.default_rng(8)
    D, dM, N, Nk, s = 3, 4, 32, 5, 2
    n = N // s
    x = np.floor(rng.uniform(0, 256, (D, N, N))).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, D, Nk, Nk)).astype(np.float32); f = rng.uniform(-1, 1, (D, dM, Nk, Nk)).astype(np.float32)
    b = rng.uniform(-1, 1, dM).astype(np.float32); p = rng.uniform(-1, 1, D).astype(np.float32)
    layers, cfreq, _ = R.autoenc_fft(x.astype(np.float64), [c.astype(np.float64), f.astype(np.float64)],
                                     [b.astype(np.float64), p.astype(np.float64)], [s, -s])
    sizes = [D * n * n, dM * n * n, D * n * n, D * N * N]
    lay = np.zeros(sum(sizes), np.float32)
    W = dM * D * n * (n // 2 + 1) * 2
    cf = np.zeros(2 * W, np.float32)
    nc = C.c_int(0)
    cc, bb, ff, pp = c.copy(), b.copy(), f.copy(), p.copy()
    wrap.w_fft_pair(_p(x), _p(lay), _p(cc), _p(bb), _p(ff), _p(pp), _p(cf), C.byref(nc), D, dM, N, Nk, s, 1, 0, C.c_float(0.2), 0)
    assert nc.value == 2
    off = 0
    for l, sz in enumerate(sizes, start=1):
        ref = layers[l].ravel()
        assert np.abs(lay[off:off + sz] - ref).max() < 1e-4 * np.abs(ref).max(), l
        off += sz
    Cs = cf[:W].view(np.complex64).reshape(dM, D, n, n // 2 + 1)
    assert np.abs(Cs - cfreq[0]).max() < 1e-5 * np.abs(cfreq[0]).max()
    # second call: cache hit (spectra loaded from net_cfreq) with fft_l = 0: only layers.back() is produced (Appendix B-6)
    lay2 = np.zeros_like(lay)
    wrap.w_fft_pair(_p(x), _p(lay2), _p(cc), _p(bb), _p(ff), _p(pp), _p(cf), C.byref(nc), D, dM, N, Nk, s, 0, 0, C.c_float(0.2), 0)
    assert np.all(lay2[:sizes[0]] == 0)
    assert np.abs(lay2[-sizes[3]:] - layers[4].ravel()).max() < 1e-4 * np.abs(layers[4]).max()
    # third call: fft_l = 1 (the burst trains on layers[1], layers[3]) then the 100-iteration backprop_fft
    # del0 = 0.01: at the reference default 0.2 the clipped (sign-like) updates make the 100-iteration trajectory chaotic --
    # the float32 replay of the oracle itself ends 0.12 (25% of the total change) away from its float64 master --
    # so a trajectory comparison is only meaningful in the smooth regime (float32 replay error 6e-6 here).
    wrap.w_fft_pair(_p(x), _p(lay2), _p(cc), _p(bb), _p(ff), _p(pp), _p(cf), C.byref(nc), D, dM, N, Nk, s, 1, 1, C.c_float(0.01), 0)
    assert np.abs(lay2[-sizes[3]:] - layers[4].ravel()).max() < 1e-4 * np.abs(layers[4]).max()
    r = R.backprop_fft(layers[1], layers[1], layers[3], cfreq[0], c.astype(np.float64), cfreq[1], f.astype(np.float64),
                       b.astype(np.float64), p.astype(np.float64), 0.01, n_iter=100)
    dw = np.abs(r["c"] - c).max()
    assert dw > 1e-3
    for a, k in ((cc, "c"), (ff, "f"), (bb, "b"), (pp, "p")):
        assert np.abs(a - r[k]).max() < 2e-5 + 1e-3 * dw, (k, np.abs(a - r[k]).max(), dw)
    Cs = cf[:W].view(np.complex64).reshape(dM, D, n, n // 2 + 1)
    assert np.abs(Cs - r["C"]).max() < 5e-3 * np.abs(r["C"] - cfreq[0]).max() + 1e-4 * np.abs(r["C"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("tied", [0, 1])
def test_vector_entry_points_spatial_mode(wrap, tied):
    rng = np.random.default_rng(9 + tied)
    dD, dM, N, Nk = 3, 4, 16, 5
    x = np.floor(rng.uniform(0, 256, (dD, N, N))).astype(np.float32)
    c = rng.uniform(-1, 1, (dM, dD, Nk, Nk)).astype(np.float32); f = rng.uniform(-1, 1, (dD, dM, Nk, Nk)).astype(np.float32)
    b = rng.uniform(-1, 1, dM).astype(np.float32); p = rng.uniform(-1, 1, dD).astype(np.float32)
    h = np.zeros((dM, N, N), np.float32)
    wrap.w_conv(_p(x), _p(h), _p(c), _p(b), dD, dM, N, N, Nk, Nk, 1)
    assert np.abs(h - S.conv(x, c, b)).max() < 1e-5 * np.abs(h).max()
    o = np.zeros((dD, N, N), np.float32)
    wrap.w_conv(_p(h), _p(o), _p(f), _p(p), dM, dD, N, N, Nk, Nk, 1)
    z = lambda a: np.zeros_like(a)
    mom = [z(c), z(b), z(f), z(p)]
    ref = S.backprop_gpu(x, o, h, c, b, f, p, *mom, 0.2, 0.9, tied=bool(tied))
    arrs = [c.copy(), b.copy(), f.copy(), p.copy(), z(c), z(b), z(f), z(p), z(c), z(b), z(f), z(p)]
    wrap.w_backprop_gpu(_p(x), _p(o), _p(h), *[_p(a) for a in arrs], C.c_float(0.2), C.c_float(0.9), tied, dD, dM, N, N, Nk, Nk)
    names = ["c", "b", "f", "p", "dc", "db", "df", "dp", "ddc", "ddb", "ddf", "ddp"]
    for a, r, k in zip(arrs, ref, names):
        if r is None or (tied and k in ("df", "ddf")):
            continue
        sc = max(np.abs(ref[4]).max(), 1e-6) if not k.startswith("dd") else max(np.abs(r).max(), 1e-9)
        assert np.abs(a - r).max() < 1e-6 + 1e-3 * sc, k
