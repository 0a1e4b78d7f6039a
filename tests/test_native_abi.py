"""The C-ABI shared library: it loads, it exports every symbol include/kiez_amd.h declares, the ctypes
prototypes cover them all, and without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "kiez_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from kiez_amd import _native as N
    lib = N.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libkiez_amd.so does not export {name}"
    bound = {s[0] for s in N.SYMBOLS}
    assert set(declared) == bound, f"ctypes prototypes out of sync with the header: {set(declared) ^ bound}"
    assert lib.kz_abi_version() == N.ABI_VERSION == 7


def test_code_object_is_gfx950_only():
    import re
    so = (ROOT / "kiez_amd" / "libkiez_amd.so").read_bytes()
    # the offload bundle's entry ids name the targets of the embedded code objects
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", so))
    assert targets == {b"gfx950"}, targets
    assert b"sm_90" not in so and b"nvptx" not in so
    assert b"rocprim" not in so.lower() and b"hipcub" not in so.lower()     # every device kernel of the library is its own


def test_no_gpu_means_loud_failure(have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    from kiez_amd import Kiez
    with pytest.raises(RuntimeError, match="no MI355X"):
        Kiez().fit(np.zeros((4, 3)), np.zeros((4, 3)))


def test_product_package_never_imports_the_oracle_or_a_cpu_math_library():
    """The oracle is test infrastructure; the product path must not route through it or any CPU fallback."""
    for f in (ROOT / "kiez_amd").glob("*.py"):
        for line in f.read_text().splitlines():
            s = line.strip()
            if s.startswith("import ") or s.startswith("from "):
                assert "oracle" not in s, f"{f}: product code imports the oracle ({s})"
                assert "sklearn" not in s and "scipy" not in s, f"{f}: product path must not compute on the CPU ({s})"


@pytest.mark.gpu
def test_context_and_error_reporting():
    from kiez_amd import _native as N
    ctx = N.Context.get()
    with pytest.raises(ValueError, match="unknown option"):
        ctx.set_option("nope", 1)
    a = ctx.to_device(np.arange(12, dtype=np.float64).reshape(3, 4))
    np.testing.assert_array_equal(a.numpy(), np.arange(12, dtype=np.float64).reshape(3, 4))
    m = N.DeviceMatrix(ctx, np.random.rand(10, 4), "euclidean")
    with pytest.raises(ValueError, match="Expected n_neighbors"):
        N.knn(ctx, m, m, 10, exclude_self=True)
    big = N.DeviceMatrix(ctx, np.random.rand(300, 4), "euclidean")
    d, i, st = N.knn(ctx, big, big, 200)          # beyond the fused kernels' 110: the exact float64 route, any k <= n
    assert i.shape == (300, 200) and st["n_fallback_rows"] == 300
    with pytest.raises(NotImplementedError, match="maximum"):
        huge = N.DeviceMatrix(ctx, np.random.rand(5000, 4), "euclidean")
        N.knn(ctx, huge, huge, 4500)


def test_public_options_are_the_four_the_header_documents():
    """include/kiez_amd.h documents four options; kz_options.h (the one table behind kz_ctx_set_option) flags exactly those
    KZ_OPT_PUBLIC.  Everything else is an internal knob outside the ABI's promise."""
    import re
    table = (ROOT / "kiez_amd" / "csrc" / "kz_options.h").read_text()
    rows = re.findall(r'\{"(\w+)", KZ_OPT_(?:INT|BOOL|F64), KZ_O\(\w+\), [^,]+, [^,]+, [^,]+, (KZ_OPT_PUBLIC|KZ_OPT_SET|0)', table)
    public = {n for n, f in rows if f == "KZ_OPT_PUBLIC"}
    assert public == {"precision", "dual_stride", "dual_max_gb", "eps_scale"} and len(rows) >= 30
    header = (ROOT / "include" / "kiez_amd.h").read_text()
    block = header[header.index("/* Options (the public contract"):header.index("int kz_ctx_set_option")]
    assert set(re.findall(r'^ \*   "(\w+)"', block, flags=re.M)) == public
    assert block.count("\n") <= 12        # (round 4: 28 lines for 34 names)
