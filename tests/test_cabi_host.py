"""CPU checks of the boundary: the library loads, exports every symbol declared in include/mpsengine.h,
the ctypes structures match the C layout, and the product refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from renormalizer_amd import engine as E

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "mpsengine.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpse_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert len(decl) >= 30
    assert decl == E.EXPORTED_SYMBOLS, (set(decl) ^ set(E.EXPORTED_SYMBOLS))


def test_library_exports_all_symbols():
    if not os.path.exists(E.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(E.LIB_PATH)          # loads without a GPU: no compute is called here
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    lib.mpse_version.restype = C.c_char_p
    assert b"gfx950" in lib.mpse_version()


def test_struct_layouts_match_c(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "mpsengine.h"\nint main(){printf("%zu %zu %zu %zu\\n", '
                   "sizeof(mpse_index), sizeof(mpse_gemm_desc), sizeof(mpse_dims), sizeof(mpse_heff));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(E.mpse_index), C.sizeof(E.mpse_gemm_desc), C.sizeof(E.mpse_dims), C.sizeof(E.mpse_heff)]


def test_no_cpu_fallback():
    """Without a usable HIP device the engine must fail loudly instead of computing on the host."""
    lib = E.load_library()
    p = C.c_void_p()
    st = lib.mpse_ctx_create(0, C.byref(p))
    if st == 0:                      # a GPU is present (GPU box): nothing to check here
        lib.mpse_ctx_destroy(p)
        pytest.skip("GPU present")
    with pytest.raises(E.EngineError):
        E.Engine()
    import renormalizer_amd
    src = open(os.path.join(os.path.dirname(renormalizer_amd.__file__), "engine.py")).read()
    assert "oracle" not in src.replace("``oracle/``", "")   # the product never imports the test oracle


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, "renormalizer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), os.path.join(root, f)


def test_host_qn_and_selection_logic():
    from renormalizer_amd.mps.svd_qn import add_outer, get_qn_mask, qn_blocks
    from renormalizer_amd.mps.basis_select import select_basis_indices
    from oracle import mps_oracle as orc
    rng = np.random.default_rng(3)
    qnl, sig, qnr = rng.integers(0, 3, (7, 2)), rng.integers(0, 2, (3, 2)), rng.integers(0, 3, (5, 2))
    assert np.array_equal(add_outer(qnl, sig), orc.add_outer(qnl, sig))
    big = add_outer(add_outer(qnl, sig), qnr)
    assert big.shape == (7, 3, 5, 2)
    assert np.array_equal(get_qn_mask(big, [2, 2]), orc.get_qn_mask(big, [2, 2]))
    mine = qn_blocks(add_outer(qnl, sig), qnr, np.array([2, 2]))
    ref = orc.qn_blocks(add_outer(qnl, sig), qnr, np.array([2, 2]))
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        assert tuple(a[0]) == b[0] and tuple(a[1]) == b[1]
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    s = rng.random(30)
    s[4] = s[9]
    qn = rng.integers(0, 3, (30, 1)).tolist()
    for percent in (0, 0.2, 1.0):
        for mmax in (5, 17, 100):
            assert select_basis_indices(s, qn, mmax, percent) == orc.select_basis_indices(s, qn, mmax, percent)
