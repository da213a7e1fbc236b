#!/usr/bin/env python3
"""Software-managed MFMA hazards the compiler cannot see through inline asm (gfx950).

The hazard recognizer inserts the wait states the ISA requires around MFMAs only for instructions it knows to be VALU; a VALU
instruction inside an `asm` statement is opaque to it.  Two cases matter for these kernels:

  (1) asm VALU writes a VGPR, an MFMA reads it (SrcA/B/C) fewer than 2 wait states later
      (this broke the W = 128 / one-plane bf16 kernel: v_cvt_pk_bf16_f32 directly in front of the MFMA; the compiler puts
      `s_nop 1` there when the conversion is its own instruction);
  (2) an MFMA writes a VGPR/AGPR, an asm VALU reads or overwrites it fewer than passes + 3 wait states later.

Wait states are counted the way the compiler counts them: one per instruction, N + 1 for `s_nop N`; labels and branches do not
help (a join point may be reached from the shorter path).  Usage:  isa_hazards.py file.s [...]   (exit 1 on a finding)
"""
import re, sys

PASSES = {"v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x2f32": 16, "v_mfma_f32_32x32x16_bf16": 8}
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(op):
    out = set()
    for m in REG.finditer(op):
        lo = int(m.group(2) if m.group(2) is not None else m.group(4))
        hi = int(m.group(3) if m.group(3) is not None else m.group(4))
        out |= {(m.group(1), r) for r in range(lo, hi + 1)}
    return out


def parse(path):
    """-> {kernel: [(lineno, mnemonic, [operands], in_asm)]}"""
    kernels, cur, in_asm = {}, None, False
    for n, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.lstrip().startswith(";;#ASMSTART"):
            in_asm = True; continue
        if raw.lstrip().startswith(";;#ASMEND"):
            in_asm = False; continue
        m = re.match(r"^(_Z\w+|\w+_k\w*):", raw)
        if m and not raw.startswith("."):
            cur = kernels.setdefault(m.group(1), []); continue
        if raw.startswith(".Lfunc_end"):
            cur = None; continue
        if cur is None or not line or line.startswith(".") or line.endswith(":"):
            continue
        parts = line.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((n, mn, ops, in_asm))
    return kernels


def waits(ins):
    _, mn, ops, _ = ins
    if mn == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def check(path):
    findings = []
    for name, code in parse(path).items():
        for i, (n, mn, ops, in_asm) in enumerate(code):
            if mn.startswith("v_mfma"):
                src = set().union(*(regs(o) for o in ops[1:4])) if len(ops) >= 4 else set()
                w, j = 0, i - 1
                while j >= 0 and w < 2:
                    pn, pmn, pops, pasm = code[j]
                    if pasm and pmn.startswith("v_") and pops and regs(pops[0]) & src:
                        findings.append(f"{path}:{n}: {name[:60]}: {mn} reads {pops[0]} written by asm `{pmn}` (line {pn}) "
                                        f"{w} wait state(s) earlier; 2 required")
                    w += waits(code[j]); j -= 1
                need = PASSES.get(mn, 16) + 3
                dst = regs(ops[0]) if ops else set()
                w, j = 0, i + 1
                while j < len(code) and w < need:
                    fn, fmn, fops, fasm = code[j]
                    if fasm and fmn.startswith("v_") and any(regs(o) & dst for o in fops):
                        findings.append(f"{path}:{fn}: {name[:60]}: asm `{fmn}` touches {ops[0]} written by {mn} (line {n}) "
                                        f"{w} wait state(s) earlier; {need} required")
                    # a later MFMA on the same accumulator serialises behind this one in hardware; stop at it
                    if fmn.startswith("v_mfma") and fops and regs(fops[0]) & dst:
                        break
                    w += waits(code[j]); j += 1
    return findings


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        bad += check(p)
    for b in bad:
        print(b)
    print(f"{len(bad)} hazard finding(s) in {len(sys.argv) - 1} file(s)")
    sys.exit(1 if bad else 0)
