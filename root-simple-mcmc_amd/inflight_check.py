"""Static check of hand-placed LDS reads in a `hipcc -S` / `-save-temps` device listing (used by build.py on every
step-kernel translation unit, and by tools/check_inflight_regs.py).

The step kernels read the decomposition pieces, the normal-transform tables and the fold's operands with
inline-assembly `ds_read_*` and wait for them with their own `s_waitcnt lgkmcnt(N)`.  Between such a read and the wait
that covers it the destination registers do not hold the value yet, while the compiler -- which sees an ordinarily
defined value -- is free to copy or spill them.  It did: in the 31-dimension family `v_accvgpr_write` of the registers
stood right behind the reads of the fold's operand prefetch, and the pooled moments were wrong.  No source-level
construct forbids that, so the listing is checked instead: the build fails if a compiler instruction touches a
register with an assembly read in flight.

The walk is linear (block layout order, no control-flow analysis): the reads live in the straight-line body of the
step loop.  LDS operations return in issue order, so a wait with count N completes all but the N youngest of them,
the compiler's own included.
"""
import re


def kernels(lines, want=""):
    out, name, start = [], None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN5smcmc11step_kernel\S*):", l)
        if m:
            name, start = m.group(1), i
        elif name and l.startswith(".Lfunc_end"):
            if want in name:
                out.append((name, lines[start:i]))
            name = None
    return out


def vregs(args):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", args):
        regs |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", args):
        regs.add(int(a))
    return regs


def check(body):
    """(number of assembly reads, [(line, text, registers)]) for one kernel's lines."""
    inflight = {}            # register -> serial number (over all LDS operations) of the assembly read that writes it
    serial, findings, reads = 0, [], 0
    in_asm = False
    for idx, l in enumerate(body):
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", l)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        wait = re.search(r"lgkmcnt\((\d+)\)", args) if op == "s_waitcnt" else None
        if in_asm and op.startswith("ds_read"):
            serial += 1
            reads += 1
            for r in vregs(args.split(",")[0]):
                inflight[r] = serial
            continue
        if wait:
            n = int(wait.group(1))
            inflight = {r: s for r, s in inflight.items() if s > serial - n}
            continue
        if op.startswith("ds_"):
            serial += 1
        if in_asm:
            continue
        hit = vregs(args) & set(inflight)
        if hit:
            findings.append((idx, l.strip(), sorted(hit)))
    return reads, findings


def check_listing(path, want=""):
    """[(kernel name, assembly reads, findings)] for every step_kernel of the listing."""
    lines = open(path).read().splitlines()
    return [(name,) + check(body) for name, body in kernels(lines, want)]
