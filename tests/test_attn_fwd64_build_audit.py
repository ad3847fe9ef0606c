"""attn_fwd64.hip owns the accumulator file (and VGPRs v192-v255) by register NUMBER inside its asm statements.  A compiler spill into
the accumulator file, or any scratch use, would corrupt results silently (no fault, possibly only on some inputs), so every build is
audited: no v_accvgpr_* / AGPR operand and no scratch instruction outside the asm statements (tools/attn_fwd64_audit.py).  CPU-only:
hipcc cross-compiles gfx950 without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_attn_fwd64_build_has_no_compiler_agpr_use_and_no_scratch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_fwd64_audit.py")], capture_output=True, text=True, cwd=ROOT,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "compiler AGPR uses outside asm: 0; scratch instructions: 0" in r.stdout, r.stdout
