"""The C-ABI library loads and exports every symbol include/lmono_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "lmono_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lmono_[a-z0-9_]+)\s*\(", txt)))


def test_build_produces_library():
    import __graft_entry__ as g
    g.build()
    import lmono_amd
    assert os.path.exists(lmono_amd.lib_path())


def test_every_declared_symbol_is_exported():
    import lmono_amd
    from lmono_amd import capi
    lib = ctypes.CDLL(lmono_amd.lib_path())
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(capi.SYMBOLS) == names, (sorted(set(names) - set(capi.SYMBOLS)), sorted(set(capi.SYMBOLS) - set(names)))
    assert b"gfx950" in ctypes.cast(lib.lmono_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_no_cpu_fallback_without_gpu():
    """The product path fails loudly when no GPU is present: lmono_create returns NULL -> LmonoError."""
    import pytest
    import torch
    import lmono_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lmono_amd.LmonoError):
        lmono_amd.Context(0)


def test_product_does_not_import_oracle():
    """Nothing under lmono_amd/ may reference oracle/ (the oracle is the checker, never the product)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "lmono_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle/" not in txt.replace("CPU oracle", "") or "#include" not in txt.split("oracle/")[0][-40:], f
                assert "from oracle" not in txt and "import oracle" not in txt and "liblmono_oracle" not in txt, f


def test_bench_touches_oracle_only_in_cpu_baseline_leg():
    """bench.py may use oracle/ only as the cpu_baseline (and the parity figures reported with it); inputs come from
    workloads/, which must not depend on the oracle either."""
    src = open(os.path.join(ROOT, "bench.py")).read().split("\n")
    for k, line in enumerate(src):
        if "from oracle" in line or "import oracle" in line:
            assert "cpu_baseline leg" in src[k - 1], "oracle import outside the cpu_baseline leg: line %d" % (k + 1)
    for f in os.listdir(os.path.join(ROOT, "workloads")):
        if f.endswith((".py", ".c", ".h")):
            txt = open(os.path.join(ROOT, "workloads", f)).read()
            assert "from oracle" not in txt and "import oracle" not in txt and "lo_oracle.h" not in txt, f
