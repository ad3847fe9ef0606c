"""CPU: the C-ABI library loads and exports every symbol include/mantis_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mantis_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(mantis_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from mantis_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mantis_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding table and header disagree"


def test_no_oracle_import_in_product():
    """The product package must not reach the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "mantis_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn
