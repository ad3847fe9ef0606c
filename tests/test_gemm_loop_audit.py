"""Build audit of the ring GEMM kernels' K loops (tools/gemm_loop_audit.py): no register-allocator traffic (`v_accvgpr_*`, `scratch_*`)
and no stray global accesses between the MFMAs of any ring / ring16 instantiation.  A change elsewhere in the kernel can put them
there silently (round 4: a K-split reduction variant cost +11 % on every NN launch this way); parity tests cannot see it."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("MANTIS_SKIP_BUILD_AUDIT") == "1", reason="MANTIS_SKIP_BUILD_AUDIT=1")
def test_ring_gemm_k_loops_are_clean():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gemm_loop_audit as A
    asm = A.compile_asm(os.path.join(ROOT, "mantis_amd", "csrc", "gemm.hip"))
    report, bad = A.audit(asm)
    assert len(report) >= 19, f"only {len(report)} ring kernels found in the assembly"
    dirty = [(n, h[:4]) for n, _, h in report if h]
    assert not dirty and bad == 0, dirty
    # every ring16 loop carries exactly its 64 (8 waves) or 128 (4 waves) MFMAs per K-step
    for name, what, _ in report:
        if "ring16" in name:
            assert what.endswith("64 MFMAs") or what.endswith("128 MFMAs"), (name, what)


@pytest.mark.skipif(os.environ.get("MANTIS_SKIP_BUILD_AUDIT") == "1", reason="MANTIS_SKIP_BUILD_AUDIT=1")
def test_ring176_gemm_k_loops_are_clean():
    """csrc/gemm176.hip (round 6): every instantiation of the 176 x 256 kernel carries exactly its 88 MFMAs per K-step and nothing the
    schedule did not place there; no spills (the kernel runs one wave per SIMD on 176 AGPRs + ~150 VGPRs)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gemm_loop_audit as A
    asm = A.compile_asm(os.path.join(ROOT, "mantis_amd", "csrc", "gemm176.hip"))
    report, bad = A.audit(asm)
    assert len(report) >= 5 and bad == 0, [(n, w, h[:4]) for n, w, h in report if h or "not found" in w]
    for name, what, _ in report:
        assert what.endswith("88 MFMAs"), (name, what)
    text = open(asm).read()
    import re
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)] + [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)]
    assert spills and max(spills) == 0, "the 176-row kernel spills"


@pytest.mark.skipif(os.environ.get("MANTIS_SKIP_BUILD_AUDIT") == "1", reason="MANTIS_SKIP_BUILD_AUDIT=1")
def test_fp8_ring_gemm_k_loops_are_clean():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gemm_loop_audit as A
    asm = A.compile_asm(os.path.join(ROOT, "mantis_amd", "csrc", "gemm_fp8.hip"))
    report, bad = A.audit(asm)
    assert len(report) >= 4 and bad == 0, [(n, h[:4]) for n, _, h in report if h]


@pytest.mark.skipif(os.environ.get("MANTIS_SKIP_BUILD_AUDIT") == "1", reason="MANTIS_SKIP_BUILD_AUDIT=1")
def test_attn_dq64_owns_its_accumulator_file():
    """csrc/attn_dq64.hip names AGPRs by number inside inline asm: the compiler must not touch the accumulator file, spill, or drain
    the kernel's four-tile LDS ring with a vmcnt(0) of its own (tools/attn_dq64_audit.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import attn_dq64_audit as A
    problems = A.audit(A.assembly())
    assert not problems, problems
