#!/usr/bin/env python3
"""Build audit of hand-issued ASYNCHRONOUS LDS reads (round-5 advisor finding, csrc/attn.hip dK/dV group kernel; the ring GEMM kernels use the
same idiom).  A fragment is requested by one asm statement (`ds_read_b64_tr_b16` / `ds_read_b128` with an "=v" output) and waited for by a
LATER one (`s_waitcnt lgkmcnt(N)`).  The compiler believes the destination is defined as soon as the first statement has issued: any
instruction it places between the two that reads or writes that register -- a `v_mov` forming a tuple, a `v_accvgpr_write`, a spill -- sees
the register before the LDS data has landed, with no hardware interlock.  Whether the destination registers stay untouched depends on
register allocation, i.e. on the toolchain; parity tests only see it when it has already gone wrong.

This tool compiles a source to assembly with the product's flags and replays every kernel's instruction stream: LDS operations retire in
order, so a read is outstanding until an `s_waitcnt lgkmcnt(N)` leaves at most N younger LGKM operations; while a read is outstanding, no other
instruction may name a register of its destination.  (Straight-line tracking: the state is dropped at labels and branches -- the audited
reads live inside unrolled steps.)  Scalar memory operations share the counter and return out of order: one outstanding `s_load` /
`s_buffer_load` makes every counted wait meaningless, which the tool reports as well.

    python tools/lds_async_read_audit.py [attn|gemm|gemm176|path.hip|path.s] [kernel-name-regex]      exit 1 + report on a violation"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

READS = ("ds_read_b64_tr_b16", "ds_read_b128", "ds_read_b64", "ds_read2", "ds_read_b32", "ds_read_u", "ds_read_i", "ds_read_b96")
LGKM_OTHER = ("ds_write", "ds_add", "ds_swizzle", "ds_bpermute", "ds_permute", "ds_append", "ds_consume", "ds_max", "ds_min", "ds_or", "ds_and",
              "s_sendmsg", "s_getreg")          # (s_getreg does not count; harmless here)
SMEM = ("s_load_", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")


def compile_asm(name):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gemm_loop_audit
    src = name if os.path.exists(name) else os.path.join(ROOT, "mantis_amd", "csrc", name + ".hip")
    return gemm_loop_audit.compile_asm(src)


def vregs(code):
    """set of VGPR numbers named by an instruction's operands"""
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", code):
        regs.add(int(m.group(1)))
    return regs


def first_operand_regs(code):
    ops = code.split(None, 1)[1] if " " in code.strip() else ""
    return vregs(ops.split(",")[0])


def audit(asm_path, kernel_re=None, only_tr=False):
    """[(kernel, line number, instruction, offending registers, the read they belong to)] and per-kernel counts of audited reads"""
    viol, counts = [], {}
    kernel, pend, smem_out = None, [], 0       # pend: outstanding LGKM ops, oldest first: (dst regs or None, text)
    for n, line in enumerate(open(asm_path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kernel = m.group(1) if (kernel_re is None or re.search(kernel_re, m.group(1))) else None
            pend, smem_out = [], 0
            continue
        if kernel is None or not t or t.startswith((";", ".", "//")):
            if re.match(r"^\.LBB\d+_\d+:", t):
                pend, smem_out = [], 0
            continue
        code = t.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        if op == "s_endpgm":
            kernel = None
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
            pend, smem_out = [], 0
            continue
        if op == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", code)
            if mm:
                keep = int(mm.group(1))
                if smem_out and keep > 0 and any(p[0] for p in pend):
                    viol.append((kernel, n, code, set(), "counted lgkmcnt wait with a scalar memory operation outstanding"))
                pend = pend[len(pend) - keep:] if keep else []
                if keep == 0:
                    smem_out = 0
            continue
        touched = vregs(code)
        for dst, what in pend:
            if dst and (touched & dst):
                viol.append((kernel, n, code, touched & dst, what))
        if op.startswith(READS):
            dst = first_operand_regs(code)
            audited = op.startswith("ds_read_b64_tr_b16") or not only_tr
            pend.append((dst if audited else None, f"line {n}: {code}"))
            if audited:
                counts[kernel] = counts.get(kernel, 0) + 1
        elif op.startswith("ds_") or op.startswith(LGKM_OTHER):
            pend.append((None, code))
        elif op.startswith(SMEM):
            smem_out += 1
    return viol, counts


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "attn"
    kre = sys.argv[2] if len(sys.argv) > 2 else None
    asm = what if what.endswith(".s") else compile_asm(what)
    viol, counts = audit(asm, kre)
    for k, c in sorted(counts.items()):
        dn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        nv = sum(1 for v in viol if v[0] == k)
        print(f"{'BAD ' if nv else 'ok  '} {dn}: {c} LDS reads tracked, {nv} violations")
    for k, n, code, regs, rd in viol[:20]:
        print(f"  line {n}: `{code}` names v{sorted(regs)} while outstanding: {rd}")
    print(f"{len(counts)} kernels audited, {len(viol)} violations")
    return 1 if viol or not counts else 0


if __name__ == "__main__":
    sys.exit(main())
