"""CPU tests of the drop-in boundary: libcsdr_hip.so loads without a GPU, exports every symbol include/csdr_hip.h
declares, the ctypes binding covers them all, and the product fails loudly (no CPU fallback) when no device exists."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "csdr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(csdr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from cubicsdr_amd import build, hip
    build.build(verbose=False)
    return hip.lib()


def test_header_symbols_are_exported(lib):
    names = declared_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libcsdr_hip.so does not export %s" % n


def test_binding_covers_the_header():
    from cubicsdr_amd import hip
    assert sorted(hip.ABI) == declared_functions()


def test_struct_layouts_match_the_header():
    from cubicsdr_amd import hip
    assert C.sizeof(hip.DemodParams) == 24
    assert C.sizeof(hip.BlockResult) == 48
    assert hip.BlockResult.level_accum.offset == 16 and hip.BlockResult.nco_theta.offset == 32


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.csdr_ctx_create(0, None, C.byref(h))
    assert rc == -3 and not h.value                       # CSDR_EHIP
    assert b"no HIP device" in lib.csdr_last_error()
    from cubicsdr_amd.engine import Context
    from cubicsdr_amd.hip import CsdrError
    with pytest.raises(CsdrError):
        Context(0)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under cubicsdr_amd/ or include/ may reference it"""
    bad = []
    for base in ("cubicsdr_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            for f in fn:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(import|from)\s+oracle|oracle/|liquid_port|liquid_ref|libliquid", txt):
                        bad.append(os.path.join(dp, f))
                    if re.search(r"tests/emu|hip_emu|libcsdr_emu|build_emu", txt):      # the host-thread emulation is test-only too
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
