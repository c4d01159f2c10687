#!/usr/bin/env python3
"""Static check of the hand-placed stream rings of perchain_step_kernel in a `hipcc -S` listing.

The kernel issues its stream loads in inline assembly (`global_load_dwordx2 ... nt`) and takes the values out with
`s_waitcnt vmcnt(N)` + `v_mov_b64` in one assembly statement: the compiler knows nothing of the loads being in flight,
so nothing but those two statements may touch a ring register while the kernel runs the stream loops --
a compiler-made copy or spill of a ring register between a load and its wait would read a register the load has not
written yet.  For every instantiation in the listing this checks that

  * every ring register is written only by the ring loads and read only inside the take statements,
  * the loads of a loop body use the registers of the priming loads in the same order (the ring does not rotate).

usage: python tools/check_ring_regs.py listing.s         (exit code 1 and a report on a violation)
"""
import re
import sys


def kernels(text):
    lines = text.splitlines()
    out, name, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN5smcmc20perchain_step_kernel\S*):", l)
        if m:
            name, start = m.group(1), i
        elif name and l.startswith(".Lfunc_end"):
            out[name] = lines[start:i]
            name = None
    return out


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def vregs(line):
    found = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        found |= regs_of(tok)
    return found


def check(name, body):
    """Sections run from the first ring load after a drain (a bare `s_waitcnt vmcnt(0)` statement) to the next drain, in
    listing order -- each stream loop of the kernel is laid out contiguously between its priming loads and its drain."""
    stmts = []             # (first line, last line, [instructions]) of every inline-assembly statement
    cur, first = None, 0
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            cur, first = [], i
        elif "#ASMEND" in l and cur is not None:
            stmts.append((first, i, cur))
            cur = None
        elif cur is not None:
            cur.append(l.strip())
    sections, start, loads = [], None, []
    for a, b, ins in stmts:
        m = [re.match(r"global_load_dwordx2 (v\[\d+:\d+\]), .* nt", x) for x in ins]
        if any(m):
            if start is None:
                start, loads = a, []
            loads += [x.group(1) for x in m if x]
        elif ins == ["s_waitcnt vmcnt(0)"] and start is not None:
            sections.append((start, b, loads))
            start = None
    if start is not None:
        return ["a stream section without its drain"]
    if not sections:
        return ["no ring loads found"]
    problems, report = [], []
    for a, b, loads in sections:
        ring = set()
        for r in loads:
            ring |= regs_of(r)
        in_asm = False
        for i in range(a, b + 1):
            l = body[i]
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif not in_asm and re.match(r"\s+[a-z]", l):
                hit = vregs(l) & ring
                if hit:
                    problems.append(f"line {i}: the compiler touches ring register(s) {sorted(hit)}: {l.strip()}")
        # the ring does not rotate: the k-th load of every later round uses the register of the k-th priming load
        period = len(dict.fromkeys(loads))
        for k, r in enumerate(loads):
            if r != loads[k % period]:
                problems.append(f"ring rotates: load {k} goes to {r}, priming load {k % period} went to {loads[k % period]}")
                break
        report.append((len(ring) // 2, len(loads)))
    return problems, report


def main():
    text = open(sys.argv[1]).read()
    ks = kernels(text)
    if not ks:
        print("no perchain_step_kernel in the listing")
        return 1
    bad = 0
    for name, body in sorted(ks.items()):
        res = check(name, body)
        if isinstance(res, list):
            print(name, res)
            bad += 1
            continue
        problems, report = res
        print(f"{name}: sections (ring size, loads in the listing) {report}, {len(problems)} problem(s)")
        for p in problems[:20]:
            print("   ", p)
        bad += 1 if problems else 0
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
