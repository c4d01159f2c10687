#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (or an llvm-objdump -d disassembly): totals by class and the
waits, for before / after comparisons of the headline kernel.
usage: python tools/isa_mix.py listing.s 'step_kernelILi50ELi0ELb1ELb0ELb1ELb0E' [more listings ...]"""
import collections
import re
import sys


def kernel_lines(path, name):
    text = open(path).read().splitlines()
    out, inside = [], False
    for l in text:
        if re.match(r"^(_Z\S*%s\S*):" % re.escape(name), l) or re.match(r"^[0-9a-f]+ <_Z\S*%s\S*>:" % re.escape(name), l):
            inside = True
            continue
        if inside:
            if l.startswith(".Lfunc_end") or re.match(r"^[0-9a-f]+ <_Z", l) or "s_endpgm" in l:
                break
            out.append(l)
    return out


def cls(n):
    if n.startswith("v_mfma"): return "mfma"
    if n.startswith("ds_"): return "lds"
    if n.startswith("s_waitcnt"): return "waitcnt"
    if n.startswith("s_nop"): return "s_nop"
    if n.startswith("s_"): return "salu"
    if n.startswith("v_accvgpr"): return "accvgpr"
    if n.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vmem_" + n.split("_")[0]
    if "f64" in n: return "valu_f64"
    if n.startswith("v_"): return "valu_other"
    return "other"


def main():
    name = sys.argv[2]
    for path in [sys.argv[1]] + sys.argv[3:]:
        ins = []
        for l in kernel_lines(path, name):
            m = re.match(r"\s+([a-z_0-9]+)(\s|$)", l)
            if m and not l.strip().startswith((";", "//", ".")):
                ins.append(m.group(1))
        c = collections.Counter(cls(i) for i in ins)
        print(path, "total", len(ins), dict(sorted(c.items())))
        d = collections.Counter(ins)
        print("   ", [(k, v) for k, v in d.most_common(14)])


if __name__ == "__main__":
    main()
