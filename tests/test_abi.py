"""The C-ABI library builds, loads, exports every symbol include/indelminer_amd.h
declares, and refuses loudly to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "indelminer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(im_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from indelminer_amd import build
    lib = build.build()
    L = C.CDLL(lib)
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "missing export: " + s
    version = int(re.search(r"#define IM_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "indelminer_amd.h")).read()).group(1))
    assert L.im_abi_version() == version == 2


def test_result_record_layout():
    from indelminer_amd import capi
    assert C.sizeof(capi.ReadResult) == 512
    assert capi.RESULT_DTYPE.itemsize == 512
    assert capi.RESULT_DTYPE.fields["ops"][1] == C.sizeof(capi.ReadResult) - 4 * capi.MAX_OPS


def test_no_silent_cpu_fallback():
    """Without a gfx950 device the context cannot be created; nothing computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from indelminer_amd import capi
    with pytest.raises(capi.IMError) as ei:
        capi.Context(0)
    assert ei.value.code == capi.E_NOGPU


def test_product_does_not_import_oracle():
    """Nothing under indelminer_amd/ may reference oracle/ (the oracle is test infrastructure)."""
    for dp, _dn, fn in os.walk(os.path.join(ROOT, "indelminer_amd")):
        for f in fn:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), (dp, f)
