#!/usr/bin/env python3
"""The loss-folded compositing kernels publish a per-workgroup partial and then take a ticket (csrc/composite.hip).  The
partial is a write-through store (`... sc1`); the ticket is a device-scope atomic add.  What makes the scheme correct is that the
store has COMPLETED before the atomic is issued: an `s_waitcnt vmcnt(0)` between the two.  A workgroup-scope release fence
compiles to nothing on gfx950, so this is checked in the ISA itself.

    isa_ticket_check.py file.s [...]      (exit 1 on a finding)

check(path) returns a list of findings; every `global_store_dwordx2 ... sc1` (the partial) must be followed by an
`s_waitcnt` with vmcnt(0) before the next `global_atomic_add`, and every kernel that takes tickets must contain one."""
import re
import sys


def check(path, need_sites=1):
    found, sites = [], 0
    name, pending = None, None
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", t)
        if m and not t.startswith(".L"):
            name, pending = m.group(1), None
            continue
        if t.startswith("global_store_dwordx2") and " sc1" in t:
            pending = ln
        elif t.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", t):
            if pending is not None:
                sites += 1
            pending = None
        elif t.startswith("global_atomic_add") and pending is not None:
            found.append(f"{path}:{ln}: {name}: ticket atomic issued with the partial store of line {pending} possibly in flight "
                         f"(no s_waitcnt vmcnt(0) in between)")
            pending = None
    if sites < need_sites:
        found.append(f"{path}: only {sites} publish-then-wait site(s) found, expected >= {need_sites}")
    return found


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        bad += check(p)
    print("\n".join(bad) if bad else "ticket publish order: ok")
    sys.exit(1 if bad else 0)
