"""Audit of attn_fwd64.hip's build: the kernel's asm statements own the accumulator file by register NUMBER, so the compiler must not
put anything there.  Fails if a v_accvgpr_* or an AGPR operand appears outside ;;#ASMSTART .. ;;#ASMEND, if the compiler's own code touches
v192-v255 (the V^T fragments: the clobber lists keep values from LIVING there across the asm statements, they do not stop a short-lived
temporary between two of them), or (optionally) on scratch use.

    python tools/attn_fwd64_audit.py [--allow-scratch]
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "mantis_amd", "csrc", "attn_fwd64.hip")


def main():
    sys.path.insert(0, os.path.join(HERE, ".."))
    from mantis_amd.build import FLAGS, EXTRA_FLAGS, _hipcc
    out = os.path.join(tempfile.mkdtemp(), "attn_fwd64.s")
    subprocess.run([_hipcc(), *FLAGS, *EXTRA_FLAGS.get("attn_fwd64", []), "-S", "--cuda-device-only", SRC, "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    bad, scratch, in_asm, kernel = [], 0, False, None
    hist = {}
    for n, line in enumerate(open(out), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w*attn_fwd64_kernel\w*):", t)
        if m:
            kernel = m.group(1)
        if kernel is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if "s_endpgm" in t:
            kernel = None
            continue
        if t.startswith(";") or t.startswith("."):
            continue
        if not in_asm:
            if "v_accvgpr" in t or re.search(r"\ba\[?\d+", t.split(";")[0]):
                bad.append((n, t))
            if "scratch_" in t:
                scratch += 1
            code = t.split(";")[0]
            regs = [int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", code)]
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
                regs += [int(m.group(1)), int(m.group(2))]
            if any(r >= 192 for r in regs):
                bad.append((n, "v192+ outside asm: " + t))
    print(f"compiler AGPR uses outside asm: {len(bad)}; scratch instructions: {scratch}")
    for n, t in bad[:20]:
        print(f"  {n}: {t}")
    if bad or (scratch and "--allow-scratch" not in sys.argv):
        sys.exit(1)


if __name__ == "__main__":
    main()
