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


def test_gemm_tile_heuristic_is_exposed_and_sane():
    """mantis_gemm_pick_variant is pure host arithmetic (no GPU needed): 12 = 256x256 ring kernel, 1 = 128x128 generic kernel."""
    from mantis_amd import _lib
    L = _lib.load()
    assert L.mantis_gemm_pick_variant(5624, 28672, 4096) == 12      # gate|up projection of the 8B step
    assert L.mantis_gemm_pick_variant(28672, 4096, 5624) == 12      # its weight gradient
    assert L.mantis_gemm_pick_variant(300, 200, 72) == 1            # small: generic kernel
    assert L.mantis_gemm_pick_variant(4608, 1152, 1152) == 1        # ViT out-projection: 90 big tiles would idle most CUs


def test_library_path_override(monkeypatch, tmp_path):
    """MANTIS_HIP_LIB points the loader at another build (A/B runs); a missing file fails loudly, never falls back."""
    import importlib
    import pytest
    from mantis_amd import _lib
    monkeypatch.setenv("MANTIS_HIP_LIB", str(tmp_path / "nope.so"))
    m = importlib.reload(_lib)
    try:
        with pytest.raises(ImportError):
            m.load()
    finally:
        monkeypatch.delenv("MANTIS_HIP_LIB")
        importlib.reload(_lib)
