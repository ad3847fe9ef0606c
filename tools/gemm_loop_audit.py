#!/usr/bin/env python3
"""Build audit of the ring GEMM kernels' K loops (csrc/gemm.hip).  The hand-placed loop leaves the register allocator no slack: a
change ANYWHERE in the kernel (an epilogue variant, the K-split reduction) can make it park a loop-carried value in an AGPR or in
scratch, and the loop then carries `v_accvgpr_write / read / mov` or `scratch_*` instructions between its MFMAs -- round 4 measured
+11 % on every NN launch from four `v_accvgpr_write_b32` in the loop (profiles/r04_experiments.md 5).  This tool compiles the source
to assembly with the product's flags and checks, for every ring16 / ring kernel instantiation, that the K loop (the innermost loop
holding the MFMAs) contains nothing but the instructions the schedule placed there.

    python tools/gemm_loop_audit.py [source.hip]      exit code 1 + a report if a loop is contaminated"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FORBIDDEN = ("scratch_", "v_accvgpr_write", "v_accvgpr_read", "v_accvgpr_mov", "buffer_store", "global_store", "global_load_dword")


def compile_asm(src):
    """assembly of `src` under the product's flags; cached per (source + headers + flags) digest so that the audits of one test session
    (K loops, asynchronous LDS reads) compile a translation unit once (gemm.hip takes ~3 minutes)"""
    from mantis_amd.build import FLAGS, EXTRA_FLAGS, _hipcc, _digest
    base = os.path.splitext(os.path.basename(src))[0]
    extra = EXTRA_FLAGS.get(base, [])
    try:
        dig = _digest(src, base)
    except OSError:
        dig = None
    cache = os.path.join(tempfile.gettempdir(), f"mantis_audit_{base}_{dig}.s") if dig else None
    if cache and os.path.exists(cache) and os.path.getsize(cache) > 0:
        return cache
    out = cache or os.path.join(tempfile.mkdtemp(), base + ".s")
    tmp = out + f".{os.getpid()}.tmp"
    subprocess.run([_hipcc(), *FLAGS, *extra, "-I", os.path.join(ROOT, "mantis_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", tmp], check=True,
                   stderr=subprocess.DEVNULL)
    os.replace(tmp, out)
    return out


def audit(asm_path):
    s = open(asm_path).read()
    report, bad = [], 0
    for m in re.finditer(r"^(_Z\d+gemm_(?:nt_ring(?:16|176)?|fp8_ring)_kernel\w+):", s, re.M):
        name = m.group(1)
        i, j = m.end(), s.index(".Lfunc_end", m.end())
        body = s[i:j].split("\n")
        mf = [k for k, l in enumerate(body) if "v_mfma" in l]
        if not mf:
            continue
        # the K loop: from the last label in front of the first MFMA cluster that is the target of a backward branch behind the last MFMA
        labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        loop = None
        for k in range(mf[-1], min(len(body), mf[-1] + 60)):
            t = body[k].strip()
            mm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < mf[0]:
                loop = (labels[mm.group(1)], k)
                break
        if loop is None:
            report.append((name, "K loop not found", []))
            bad += 1
            continue
        hits = [(k, body[k].split(";")[0].strip()) for k in range(loop[0], loop[1]) if body[k].strip().startswith(FORBIDDEN)]
        n_mfma = sum(1 for k in range(loop[0], loop[1]) if "v_mfma" in body[k])
        report.append((name, f"{loop[1] - loop[0]} lines, {n_mfma} MFMAs", hits))
        bad += bool(hits)
    return report, bad


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mantis_amd", "csrc", "gemm.hip")
    asm = src if src.endswith(".s") else compile_asm(src)
    report, bad = audit(asm)
    for name, what, hits in report:
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{'BAD ' if hits or 'not found' in what else 'ok  '} {dn}: {what}" + ("".join(f"\n      line {k}: {t}" for k, t in hits[:8])))
    print(f"{len(report)} kernels audited, {bad} contaminated")
    return 1 if bad or not report else 0


if __name__ == "__main__":
    sys.exit(main())
