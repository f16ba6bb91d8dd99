"""CPU-side boundary checks: libaefft.so loads and exports every symbol include/aefft.h declares
(no compute call is made -- there is no GPU here), and the Python prototype table covers them all."""
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
aefft = importlib.import_module("autoencoder-fft_amd")


def _declared():
    txt = open(os.path.join(ROOT, "include", "aefft.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aefft_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(aefft.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return aefft.lib()


def test_header_symbols_exported(built):
    names = _declared()
    assert len(names) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", aefft.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    missing = [n for n in names if n not in exported]
    assert not missing, missing


def test_python_prototypes_cover_header(built):
    assert sorted(aefft.SIGNATURES) == _declared()


def test_data_parallel_library_exports_its_header():
    """include/aefft_dp.h: libaefft_dp.so (libaefft.so + librccl) loads without a device and exports every declared symbol; the Python
    prototype table covers them"""
    if not os.path.exists(aefft.DP_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "autoencoder-fft_amd", "csrc"), "dp"])
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "aefft_dp.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(aefft_dp_[a-z0-9_]+)\s*\(", txt)))
    assert len(names) == 10 and names == sorted(aefft.DP_SIGNATURES)
    out = subprocess.run(["nm", "-D", "--defined-only", aefft.DP_LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert not [n for n in names if n not in exported]
    aefft.dp_lib()
    und = subprocess.run(["nm", "-D", "--undefined-only", aefft.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "nccl" not in und, "libaefft.so itself must not depend on RCCL"


def test_flag_table_matches_header():
    txt = open(os.path.join(ROOT, "include", "aefft.h")).read()
    hdr = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r"AEFFT_F_([A-Z]+)\s*=\s*1\s*<<\s*(\d+)", txt)}
    assert hdr == aefft.FLAGS and len(hdr) >= 16


def test_library_reads_the_environment_once():
    """development switches live in the context (aefft_ctx_set_flags); the only getenv is AEFFT_FLAGS at context creation"""
    hits = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "autoencoder-fft_amd", "csrc")):
        for fn in fns:
            if fn.endswith((".hip", ".cpp", ".h")):
                for i, line in enumerate(open(os.path.join(dp, fn), errors="ignore")):
                    if re.search(r"\bgetenv\s*\(", line):
                        hits.append((fn, i + 1))
    assert len(hits) == 1 and hits[0][0] == "aefft_capi.hip", hits


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aefft.AefftError):
        aefft.Context()
    import ctypes as C
    h = C.c_void_p()
    assert built.aefft_ctx_create(C.byref(h), 0, None, 1) == aefft.EHIP and not h.value


def test_product_does_not_touch_oracle():
    """the product path must never import / link the checkers under oracle/"""
    pkg = os.path.join(ROOT, "autoencoder-fft_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "np_ref" not in txt and "cpu_ref" not in txt and "oracle/" not in txt.replace("oracle/ ", ""), (dp, fn)


def test_unknown_name_in_AEFFT_FLAGS_fails_context_creation(built):
    """a typo in AEFFT_FLAGS must not silently run the default path: the first aefft_ctx_create returns AEFFT_EINVAL"""
    import sys
    code = ("import ctypes as C, importlib, sys; sys.path.insert(0, %r); m = importlib.import_module('autoencoder-fft_amd'); L = m.lib(); "
            "h = C.c_void_p(); print(L.aefft_ctx_create(C.byref(h), 0, None, 1))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, AEFFT_FLAGS="NOMFMA,BOGUS"))
    assert out.stdout.strip().endswith("1") and "BOGUS" in out.stderr, (out.stdout, out.stderr)


def test_hot_kernels_do_not_spill():
    """The back end's per-kernel resource tables (build/*.rsrc, written by the Makefile with -Rpass-analysis=kernel-resource-usage): no kernel of the
    training step may use scratch memory (a spilled register is a memory round trip inside loops that are bound by exactly those; a second call site of
    an inlined body once turned the tail launch's 23 us into 195), and the launches whose residency the design counts on keep their register budgets
    (tail_kernel<lean, no fused MSE> <= 80, with it <= 96, kspec_group_kernel<5,5,1> <= 128)."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "autoencoder-fft_amd", "csrc", "build", "*.rsrc"))
    if not files:
        pytest.skip("no resource tables (library built before the Makefile wrote them)")
    rows = {}
    for f in files:
        name = None
        for line in open(f):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1); rows[name] = {}
            for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vgprs", r" VGPRs: (\d+)")):
                m = re.search(pat, line)
                if m and name:
                    rows[name][key] = int(m.group(1))
    assert len(rows) > 50
    legacy = ("contract_kernelILi",)          # the scalar-FMA contraction (AEFFT_F_NOMFMA / odd shapes): 20 bytes, not on the step's path
    spills = {k: v["scratch"] for k, v in rows.items() if v.get("scratch", 0) > 0 and not any(t in k for t in legacy)}
    assert not spills, spills
    budget = {"tail_kernelILb1ELb0E": 80, "tail_kernelILb1ELb1E": 96, "kspec_group_kernelILi5ELi5ELi1E": 128, "msgrad_kernelILi8E": 128, "msgrad_kernelILi4E": 96, "wgrad_taps_kernelILi5E": 128}
    for frag, cap in budget.items():
        hit = [k for k in rows if frag in k]
        assert hit, frag
        for k in hit:
            assert rows[k]["vgprs"] <= cap, (k, rows[k])
