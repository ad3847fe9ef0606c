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


def test_async_lds_read_audit_detects_a_touched_destination(tmp_path):
    """tools/lds_async_read_audit.py on a hand-written stream: a v_mov of a hand-issued read's destination in front of its wait is a violation,
    the same move behind the wait is not, a counted wait retires exactly the older reads, and a scalar load makes a counted wait meaningless."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_async_read_audit as A
    def run(body):
        p = tmp_path / "k.s"
        p.write_text("_Z1kv:\n" + body + "\n\ts_endpgm\n")
        return A.audit(str(p))[0]
    ok = "\tds_read_b64_tr_b16 v[4:5], v1 offset:64\n\tds_read_b64_tr_b16 v[6:7], v2\n\tv_add_u32_e32 v9, v1, v2\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v8, v4"
    assert run(ok) == []
    bad = "\tds_read_b64_tr_b16 v[4:5], v1\n\tv_mov_b32_e32 v8, v5\n\ts_waitcnt lgkmcnt(0)"
    v = run(bad)
    assert len(v) == 1 and v[0][3] == {5}
    counted = "\tds_read_b64_tr_b16 v[4:5], v1\n\tds_read_b64_tr_b16 v[6:7], v1\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v8, v4\n\tv_mov_b32_e32 v9, v6\n\ts_waitcnt lgkmcnt(0)"
    v = run(counted)
    assert len(v) == 1 and v[0][3] == {6}
    smem = "\ts_load_dword s4, s[0:1], 0x0\n\tds_read_b64_tr_b16 v[4:5], v1\n\tds_read_b64_tr_b16 v[6:7], v1\n\ts_waitcnt lgkmcnt(1)"
    assert any("scalar memory" in x[4] for x in run(smem))
    accv = "\tds_read_b128 v[4:7], v1\n\tv_accvgpr_write_b32 a0, v6\n\ts_waitcnt lgkmcnt(0)"
    assert len(run(accv)) == 1


@pytest.mark.skipif(os.environ.get("MANTIS_SKIP_BUILD_AUDIT") == "1", reason="MANTIS_SKIP_BUILD_AUDIT=1")
@pytest.mark.parametrize("unit", ["attn", "gemm176", "gemm"])
def test_hand_issued_lds_reads_are_not_touched_before_their_wait(unit):
    """Round-5 advisor finding (csrc/attn.hip, dK/dV group kernel): a fragment requested by one asm statement and waited for by a later one
    must not be named by any instruction in between -- the compiler thinks it is defined at issue.  Checked on the built assembly of every
    kernel that uses the idiom (the dK/dV kernels' transposing reads, the ring GEMMs' fragment reads)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_async_read_audit as A
    viol, counts = A.audit(A.compile_asm(unit))
    assert counts, "no kernel found"
    assert not viol, [(k, n, c, sorted(r)) for k, n, c, r, _ in viol[:6]]
    if unit == "attn":
        assert any("dkv_g4" in k and c >= 500 for k, c in counts.items()), "the dK/dV group kernel's hand-issued reads were not seen"
