#!/usr/bin/env python3
"""Build audit of csrc/attn_dq64.hip (no GPU needed; tests/test_gemm_loop_audit.py runs it too).

The kernel owns the accumulator file BY REGISTER NUMBER inside its inline-asm statements (dQ accumulators a[0:127], Q fragments
a[128:191], dO fragments a[192:255]); the compiler is never told.  That is only sound while the compiler itself touches no AGPR and
spills nothing: this script compiles the file to assembly with the product's flags and fails when, for either instantiation,
  * the private segment is not empty or a `scratch_*` instruction exists (a spill),
  * more than 256 VGPRs or other than 256 AGPRs are reported (the asm clobber list must make the descriptor allocate the whole file),
  * a `v_accvgpr_*` or `v_mfma_*` instruction stands outside an `;;#ASMSTART ... ;;#ASMEND` bracket (compiler-generated),
  * the tile loop contains `s_waitcnt vmcnt(0)` (it would drain the four-tile LDS ring: the loop's only VMEM wait is the kernel's own
    `s_waitcnt vmcnt(16)`), or an `s_barrier` that the compiler preceded with such a wait.
Usage: python tools/attn_dq64_audit.py [path/to/attn_dq64.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def assembly():
    from mantis_amd.build import FLAGS, EXTRA_FLAGS, _hipcc
    src = os.path.join(ROOT, "mantis_amd", "csrc", "attn_dq64.hip")
    out = os.path.join(tempfile.mkdtemp(), "attn_dq64.s")
    flags = [f for f in FLAGS if f != "-fPIC"] + EXTRA_FLAGS.get("attn_dq64", [])
    subprocess.run([_hipcc(), *flags, "-I", os.path.join(ROOT, "mantis_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def audit(text):
    problems, kernels = [], 0
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\d+attn_bwd_dq64_kernel\w+:", l)]
    for st in starts:
        name = lines[st].split(":")[0]
        end = next(j for j in range(st, len(lines)) if lines[j].startswith(".Lfunc_end"))
        body = lines[st + 1:end]
        kernels += 1
        meta = {k: int(v) for k, v in re.findall(r"\.set " + re.escape(name) + r"\.(num_vgpr|num_agpr|private_seg_size), (\d+)", text)}
        if meta.get("private_seg_size", -1) != 0:
            problems.append(f"{name}: private segment {meta.get('private_seg_size')} bytes (spill)")
        if meta.get("num_vgpr", 999) > 256 or meta.get("num_agpr", 0) != 256:
            problems.append(f"{name}: {meta.get('num_vgpr')} VGPRs / {meta.get('num_agpr')} AGPRs (want <= 256 / == 256)")
        in_asm, mfma_idx, vm0_idx = False, [], []
        for n, l in enumerate(body):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif t.startswith("scratch_"):
                problems.append(f"{name}: {t}")
            elif (t.startswith("v_accvgpr") or t.startswith("v_mfma")) and not in_asm:
                problems.append(f"{name}: compiler-generated {t.split()[0]} outside the asm statements")
            if t.startswith("v_mfma"):
                mfma_idx.append(n)
            if re.match(r"s_waitcnt vmcnt\(0\)", t):
                vm0_idx.append(n)
        if len(mfma_idx) != 192:
            problems.append(f"{name}: {len(mfma_idx)} MFMAs (want 2 tile bodies x 96)")
        elif any(mfma_idx[0] - 160 < n < mfma_idx[-1] for n in vm0_idx):
            problems.append(f"{name}: s_waitcnt vmcnt(0) inside the tile loop (the LDS ring would be drained every tile)")
    if kernels != 2:
        problems.append(f"{kernels} attn_bwd_dq64_kernel instantiations found (want 2)")
    return problems


def main():
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else assembly()
    problems = audit(text)
    for p in problems:
        print("FAIL", p)
    print("attn_dq64 audit:", "clean" if not problems else f"{len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
